// bk_lens.cpp -- the script side of the C ABI: globe / lens loading, zoom, and the GPU lensmap
// build (emit -> hiprtc -> launch).  Host logic mirrors fisheye.c's LUA_load_globe (1752-1875),
// LUA_load_lens (1659-1750), calc_zoom (1293-1386) and create_lensmap (2367-2397).
#include <hip/hiprtc.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "bk_build_params.h"
#include "bk_emit.h"
#include "bk_internal.h"
#include "bk_lua.h"
#include "bkm.h"

using namespace bklua;
extern char **environ;

// a compiled lens module (or why there is none): what the memory cache, the disk cache or hiprtc hands back
struct CodeResult {
    int rc = BK_OK;
    std::string log;
    std::shared_ptr<std::vector<char>> code;
    bool from_disk = false;
};

static void park_compile(std::shared_future<::CodeResult> &f);      // (below, next to the code caches)

/* (int)double as the reference's x86-64 build converts it (cvttsd2si): NaN and out-of-range values become INT_MIN - where the
 * reference assigns a Lua number to an int (fisheye.c:1522, 1734, 1738) the C language leaves that case undefined */
static int h_trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

namespace bk {

struct LensProgram {
    Interp interp;
    // lens (fisheye.c:379-451 + lua_refs 328-332)
    bool lens_valid = false;
    bk_lens_info info{};
    Value lens_inverse, lens_forward, globe_plate;
    // compiled build kernels, keyed by the generated source text
    std::string module_source;
    hipModule_t module = nullptr;
    bool module_from_cache = false;      // the last compile_module() loaded its code object from BLINKY_HIP_CACHE
    hipFunction_t k_inverse = nullptr, k_corners = nullptr, k_quads = nullptr, k_resolve = nullptr, k_tiles = nullptr;
    std::shared_future<::CodeResult> pending;     // bk_set_async_compile: hiprtc running on another thread ...
    std::string pending_source;                      // ... for this generated source
    // generate_source's last answer and the interpreter activity it was given at: emitting the translation unit walks the callbacks
    // (16 us for winkel2, 43 for panini, 280 for quincuncial on the build host) and was done again on every bk_build
    struct { bool valid = false; unsigned long long activity = 0; int libm_rel = 0; bool stateless = false; std::string src, refused; } emitted;
    bool fwd_needs_host = false;  // the last forward build flagged entries for the host: the next one stops after each pass again
    std::string last_source;      // for bk_debug_kernel_source
    std::string console;          // print() output of the scripts

    explicit LensProgram(bk_ctx *ctx);
};

// ---- host versions of the converters (fisheye.c:1184-1214), on the portable libm ----------------
static void h_vector_ma(const float *a, float scale, const float *b, float *c)
{
    c[0] = a[0] + scale * b[0];
    c[1] = a[1] + scale * b[1];
    c[2] = a[2] + scale * b[2];
}
static void h_vector_normalize(float *v)
{
    float length = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    length = (float)__builtin_sqrt((double)length);
    if (length) {
        float ilength = 1 / length;
        v[0] *= ilength;
        v[1] *= ilength;
        v[2] *= ilength;
    }
}
static void h_latlon_to_ray(const MathLib &M, double lat, double lon, float *ray)
{
    double clat = M.cos(lat);
    ray[0] = (float)(M.sin(lon) * clat);
    ray[1] = (float)M.sin(lat);
    ray[2] = (float)(M.cos(lon) * clat);
}
static void h_ray_to_latlon(const MathLib &M, const float *ray, double *lat, double *lon)
{
    *lon = M.atan2((double)ray[0], (double)ray[2]);
    *lat = M.atan2((double)ray[1], M.sqrt((double)(ray[0] * ray[0] + ray[2] * ray[2])));
}

static void h_plate_uv_to_ray(const bk_plate &p, double u, double v, float *ray)
{
    u -= 0.5;
    v -= 0.5;
    v = -v;
    ray[0] = ray[1] = ray[2] = 0;
    h_vector_ma(ray, p.dist, p.forward, ray);
    h_vector_ma(ray, (float)u, p.right, ray);
    h_vector_ma(ray, (float)v, p.up, ray);
    h_vector_normalize(ray);
}
static void h_cross(const float *v1, const float *v2, float *cross)            /* mathlib.c:389 */
{
    cross[0] = v1[1] * v2[2] - v1[2] * v2[1];
    cross[1] = v1[2] * v2[0] - v1[0] * v2[2];
    cross[2] = v1[0] * v2[1] - v1[1] * v2[0];
}

static double need_num(const Values &a, size_t i, const char *fn)
{
    if (i >= a.size() || a[i].t != Value::NUM)
        throw LuaError(std::string("bad argument #") + std::to_string(i + 1) + " to '" + fn + "' (number expected)");
    return a[i].n;
}

// Host-side script execution (chunks, calc_zoom, globe loading) runs on the PLATFORM libm by
// default: that is what the reference's Lua VM calls on this machine, so lens.scale, lens_width,
// plate vectors etc. come out bit-identical to the reference's.  bk_set_host_math() switches
// to the portable bkm.h functions (= what the GPU kernels use) for platform-independent results.
LensProgram::LensProgram(bk_ctx *ctx) : interp(math_platform())
{
    interp.print_sink = [this](const std::string &s) { console += s; };          // (print's lines with their newline, io.write's text as it is)
    // the aliases init_lua installs (fisheye.c:1230-1248): cos = math.cos ... tau = math.pi*2
    static const char *alias[] = {"cos", "sin", "tan", "asin", "acos", "atan", "atan2", "sinh", "cosh", "tanh",
                                  "log", "log10", "abs", "sqrt", "exp", "pow"};
    Value math = interp.get_global("math");
    for (const char *a : alias) interp.set_global(a, math.tab()->get(Value::string(a)));
    interp.set_global("pi", math.tab()->get(Value::string("pi")));
    interp.set_global("tau", Value::number(math.tab()->get(Value::string("pi")).n * 2));
    // the three C functions scripts may call (fisheye.c:1257-1264, 1494-1537)
    interp.register_builtin("latlon_to_ray", [](Interp &I, const Values &a, Values &r) {
        float ray[3];
        h_latlon_to_ray(*I.math, need_num(a, 0, "latlon_to_ray"), need_num(a, 1, "latlon_to_ray"), ray);
        for (int i = 0; i < 3; ++i) r.push_back(Value::number((double)ray[i]));
    });
    interp.register_builtin("ray_to_latlon", [](Interp &I, const Values &a, Values &r) {
        float ray[3] = {(float)need_num(a, 0, "ray_to_latlon"), (float)need_num(a, 1, "ray_to_latlon"),
                        (float)need_num(a, 2, "ray_to_latlon")};
        double lat, lon;
        h_ray_to_latlon(*I.math, ray, &lat, &lon);
        r.push_back(Value::number(lat));
        r.push_back(Value::number(lon));
    });
    interp.register_builtin("plate_to_ray", [ctx](Interp &, const Values &a, Values &r) {
        int pi = h_trunc_to_int(need_num(a, 0, "plate_to_ray"));
        double u = need_num(a, 1, "plate_to_ray"), v = need_num(a, 2, "plate_to_ray");
        if (pi < 0 || pi >= ctx->numplates) { r.push_back(Value()); return; }       /* fisheye.c:1527-1530 */
        float ray[3];
        h_plate_uv_to_ray(ctx->plates[pi], u, v, ray);
        for (int i = 0; i < 3; ++i) r.push_back(Value::number((double)ray[i]));
    });
}

void lensprogram_free(LensProgram *p)
{
    if (!p) return;
    if (p->module) (void)hipModuleUnload(p->module);
    park_compile(p->pending);           // a compile still running must not stall the teardown
    delete p;
}

static LensProgram *prog_of(bk_ctx *ctx)
{
    if (!ctx->prog) ctx->prog = new LensProgram(ctx);
    return ctx->prog;
}

static void fill_plate(const MathLib &M, bk_plate &p, const double fwd[3], const double up[3], double fov_deg, bool *fov_ok)
{
    for (int j = 0; j < 3; ++j) p.forward[j] = (float)fwd[j];                  /* fisheye.c:1818 */
    for (int j = 0; j < 3; ++j) p.up[j] = (float)up[j];                        /* :1843 */
    h_cross(p.up, p.forward, p.right);                                          /* :1849 */
    h_cross(p.forward, p.right, p.up);                                          /* :1850 */
    p.fov = (float)(fov_deg * 3.14159265358979323846 / 180);                    /* :1858 */
    *fov_ok = p.fov > 0;                                                        /* :1861 */
    p.dist = (float)(0.5 / M.tan((double)(p.fov / 2)));                         /* :1868 */
}

}  // namespace bk

using bk::LensProgram;

// ---- globe ---------------------------------------------------------------------------------------------

extern "C" int bk_load_globe(bk_ctx *ctx, const char *src, size_t len, const char *chunkname)
{
    if (!ctx || !src) return BK_E_INVALID;
    LensProgram *P = bk::prog_of(ctx);
    Interp &I = P->interp;
    const std::string name = chunkname ? chunkname : "globe";
    // LUA_clear_globe, fisheye.c:1897-1903
    I.set_global("plates", Value());
    I.set_global("globe_plate", Value());
    ctx->numplates = 0;
    ctx->globe_valid = false;
    P->globe_plate = Value();
    try {
        I.run(std::string(src, len), name);
        Value gp = I.get_global("globe_plate");
        if (gp.is_function()) P->globe_plate = gp;                              /* :1778-1782 */
        Value plates = I.get_global("plates");
        if (plates.t != Value::TABLE || plates.tab()->length() < 1)
            return ctx->fail(BK_E_SCRIPT, "plates must be an array of one or more elements");   /* :1788 */
        // lua_next order: array part, then the rest (:1796)
        Values items;
        items.append(plates.tab()->arr.data(), plates.tab()->arr.data() + plates.tab()->arr.size());
        for (auto &kv : plates.tab()->nhash) items.push_back(kv.second);
        for (auto &kv : plates.tab()->shash) items.push_back(kv.second);
        if (items.size() > BK_MAX_PLATES)
            return ctx->fail(BK_E_SCRIPT, "globe defines %zu plates; at most %d are supported (MAX_PLATES, fisheye.c:352)",
                             items.size(), BK_MAX_PLATES);
        int i = 0;
        for (const Value &plate : items) {
            double vec[2][3];
            static const char *vname[2] = {"forward", "up"};
            if (plate.t != Value::TABLE) return ctx->fail(BK_E_SCRIPT, "plate %d: not a table", i + 1);
            for (int k = 0; k < 2; ++k) {
                Value v = plate.tab()->get(Value::number(k + 1));
                if (v.t != Value::TABLE || v.tab()->length() != 3)
                    return ctx->fail(BK_E_SCRIPT, "plate %d: %s vector is not a 3d vector", i + 1, vname[k]);   /* :1804,1829 */
                for (int j = 0; j < 3; ++j) {
                    Value e = v.tab()->get(Value::number(j + 1));
                    if (e.t != Value::NUM)
                        return ctx->fail(BK_E_SCRIPT, "plate %d: %s vector: element %d not a number", i + 1, vname[k], j + 1);
                    vec[k][j] = e.n;
                }
            }
            Value fov = plate.tab()->get(Value::number(3));
            bool fov_ok = false;
            bk::fill_plate(*I.math, ctx->plates[i], vec[0], vec[1], fov.t == Value::NUM ? fov.n : 0.0, &fov_ok);
            if (!fov_ok) return ctx->fail(BK_E_SCRIPT, "plate %d: fov must > 0", i + 1);                       /* :1863 */
            ++i;
        }
        ctx->numplates = i;                                                     /* :1872 */
        ctx->globe_valid = true;
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "could not load globe: %s", e.what());
    }
    return BK_OK;
}

extern "C" int bk_get_globe(const bk_ctx *ctx, bk_plate plates[BK_MAX_PLATES], int *n)
{
    if (!ctx) return BK_E_INVALID;
    for (int i = 0; i < ctx->numplates; ++i) plates[i] = ctx->plates[i];
    if (n) *n = ctx->numplates;
    return BK_OK;
}

extern "C" int bk_set_globe_plates(bk_ctx *ctx, const bk_plate *plates, int numplates)
{
    if (!ctx || !plates || numplates < 1 || numplates > BK_MAX_PLATES) return BK_E_INVALID;
    for (int i = 0; i < numplates; ++i) ctx->plates[i] = plates[i];
    ctx->numplates = numplates;
    ctx->globe_valid = true;
    if (ctx->prog) ctx->prog->globe_plate = Value();
    return BK_OK;
}

// ---- lens ------------------------------------------------------------------------------------------------

extern "C" int bk_load_lens(bk_ctx *ctx, const char *src, size_t len, const char *chunkname)
{
    if (!ctx || !src) return BK_E_INVALID;
    LensProgram *P = bk::prog_of(ctx);
    Interp &I = P->interp;
    const std::string name = chunkname ? chunkname : "lens";
    // LUA_clear_lens, fisheye.c:1880-1894
    for (const char *g : {"map", "max_fov", "max_vfov", "lens_width", "lens_height", "lens_inverse", "lens_forward", "onload"})
        I.set_global(g, Value());
    I.set_global("numplates", Value::number((double)ctx->numplates));
    P->lens_valid = false;
    P->lens_inverse = P->lens_forward = Value();
    memset(&P->info, 0, sizeof P->info);
    try {
        I.run(std::string(src, len), name);
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "could not load lens: %s", e.what());
    }
    bk_lens_info &info = P->info;
    info.map_type = BK_MAP_NONE;
    Value inv = I.get_global("lens_inverse"), fwd = I.get_global("lens_forward");
    if (inv.is_function()) { P->lens_inverse = inv; info.has_inverse = 1; info.map_type = BK_MAP_INVERSE; }      /* :1688-1696 */
    if (fwd.is_function()) {                                                                                  /* :1699-1709 */
        P->lens_forward = fwd;
        info.has_forward = 1;
        if (info.map_type == BK_MAP_NONE) info.map_type = BK_MAP_FORWARD;
    }
    Value map = I.get_global("map");                                                                          /* :1712-1731 */
    if (map.t == Value::STR || map.t == Value::NUM) {
        std::string fn = I.tostring(map);
        if (fn == "lens_inverse") info.map_type = BK_MAP_INVERSE;
        else if (fn == "lens_forward") info.map_type = BK_MAP_FORWARD;
        else return ctx->fail(BK_E_SCRIPT, "Unsupported map function: %s", fn.c_str());
    }
    auto num_or_zero = [&](const char *g) { Value v = I.get_global(g); return v.t == Value::NUM ? v.n : 0.0; };
    info.max_fov = h_trunc_to_int(num_or_zero("max_fov"));                                                               /* :1733-1739 */
    info.max_vfov = h_trunc_to_int(num_or_zero("max_vfov"));
    info.lens_width = num_or_zero("lens_width");                                                              /* :1741-1747 */
    info.lens_height = num_or_zero("lens_height");
    Value onload = I.get_global("onload");                                                                    /* cmd_lens :1087-1095 */
    if (onload.t == Value::STR || onload.t == Value::NUM) snprintf(info.onload, sizeof info.onload, "%s", I.tostring(onload).c_str());
    P->lens_valid = true;
    return BK_OK;
}

// "not a valid lens" / "not a valid globe" (fisheye.c:1080-1083, 1157-1160): the host could not even
// read the script; the next bk_build then clears the map and reports the invalid state
extern "C" int bk_clear_lens(bk_ctx *ctx)
{
    if (!ctx) return BK_E_INVALID;
    if (ctx->prog) { ctx->prog->lens_valid = false; ctx->prog->lens_inverse = ctx->prog->lens_forward = Value(); }
    return BK_OK;
}

extern "C" int bk_clear_globe(bk_ctx *ctx)
{
    if (!ctx) return BK_E_INVALID;
    ctx->globe_valid = false;
    ctx->numplates = 0;
    if (ctx->prog) ctx->prog->globe_plate = Value();
    return BK_OK;
}

extern "C" int bk_get_lens_info(const bk_ctx *ctx, bk_lens_info *out)
{
    if (!ctx || !out) return BK_E_INVALID;
    if (!ctx->prog || !ctx->prog->lens_valid) { memset(out, 0, sizeof *out); return BK_E_STATE; }
    *out = ctx->prog->info;
    return BK_OK;
}

// ---- callbacks on the host (calc_zoom only; per-pixel evaluation happens on the GPU) ---------------------

// LUAtoC_lens_forward, fisheye.c:1590-1632: 1 ok, 0 nil, -1 malformed
static int host_lens_forward(bk_ctx *ctx, const float ray[3], double *x, double *y)
{
    LensProgram *P = ctx->prog;
    Values r = P->interp.call(P->lens_forward, Values{Value::number((double)ray[0]), Value::number((double)ray[1]),
                                                      Value::number((double)ray[2])});
    if (r.size() == 2) {
        if (r[0].t == Value::NUM && r[1].t == Value::NUM) { *x = r[0].n; *y = r[1].n; return 1; }
        ctx->fail(BK_E_SCRIPT, "lens_forward returned a non-number value for x,y");
        return -1;
    }
    if (r.size() == 1) {
        if (r[0].t == Value::NIL) return 0;
        ctx->fail(BK_E_SCRIPT, "lens_forward returned a single non-nil value");
        return -1;
    }
    ctx->fail(BK_E_SCRIPT, "lens_forward returned %zu values instead of 2", r.size());
    return -1;
}

extern "C" int bk_calc_zoom(bk_ctx *ctx, double *scale_out)
{
    if (!ctx) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "no valid lens");
    if (ctx->W <= 0 || ctx->H <= 0) return ctx->fail(BK_E_STATE, "bk_calc_zoom: call bk_resize first");
    const bk_lens_info &L = P->info;
    double scale = -1;                                                                   /* :1296 */
    try {
        if (ctx->zoom_type == BK_ZOOM_FOV || ctx->zoom_type == BK_ZOOM_VFOV) {
            if (L.max_fov <= 0 || L.max_vfov <= 0)
                return ctx->fail(BK_E_ZOOM, "max_fov & max_vfov not specified, try \"f_cover\"");            /* :1303 */
            if (ctx->zoom_type == BK_ZOOM_FOV && ctx->zoom_fov > L.max_fov)
                return ctx->fail(BK_E_ZOOM, "fov must be less than %d", L.max_fov);                           /* :1307 */
            if (ctx->zoom_type == BK_ZOOM_VFOV && ctx->zoom_fov > L.max_vfov)
                return ctx->fail(BK_E_ZOOM, "vfov must be less than %d", L.max_vfov);                         /* :1311 */
            if (!L.has_forward)
                return ctx->fail(BK_E_ZOOM, "Please specify a forward mapping function in your script for FOV scaling");   /* :1343 */
            float ray[3];
            double x = 0, y = 0;
            const double fovr = ctx->zoom_fov * 3.14159265358979323846 / 180;                                 /* :1319 */
            if (ctx->zoom_type == BK_ZOOM_FOV) bk::h_latlon_to_ray(*P->interp.math, 0, fovr * 0.5, ray);       /* :1321 */
            else bk::h_latlon_to_ray(*P->interp.math, fovr * 0.5, 0, ray);                                     /* :1331 */
            const int st = host_lens_forward(ctx, ray, &x, &y);
            if (st == -1) return BK_E_SCRIPT;
            if (st == 0) return ctx->fail(BK_E_ZOOM, "ray_to_xy did not return a valid r value for determining FOV scale");
            scale = ctx->zoom_type == BK_ZOOM_FOV ? x / (ctx->W * 0.5) : y / (ctx->H * 0.5);                   /* :1323, 1333 */
        } else if (ctx->zoom_type == BK_ZOOM_CONTAIN || ctx->zoom_type == BK_ZOOM_COVER) {
            const double fit_w = L.lens_width / ctx->W, fit_h = L.lens_height / ctx->H;                        /* :1349-1350 */
            const bool wp = L.lens_width > 0, hp = L.lens_height > 0;
            if (!wp && hp) scale = fit_h;
            else if (wp && !hp) scale = fit_w;
            else if (!wp && !hp)
                return ctx->fail(BK_E_ZOOM, "neither lens_height nor lens_width are valid/specified.  Try f_fov instead.");
            else {
                const double lens_aspect = L.lens_width / L.lens_height;
                const double screen_aspect = (double)ctx->W / ctx->H;
                const bool lens_wider = lens_aspect > screen_aspect;
                if (ctx->zoom_type == BK_ZOOM_CONTAIN) scale = lens_wider ? fit_w : fit_h;
                else scale = lens_wider ? fit_h : fit_w;
            }
        }
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "lens_forward failed: %s", e.what());
    }
    // (as the reference writes it: a NaN scale - a lens_forward that returns NaN - is NOT rejected and builds an empty map)
    if (scale <= 0) return ctx->fail(BK_E_ZOOM, "init returned a scale of %f, which is  <= 0", scale);          /* :1380 */
    ctx->scale = scale;
    if (scale_out) *scale_out = scale;
    return BK_OK;
}

// ---- hiprtc ---------------------------------------------------------------------------------------------

// Optional on-disk cache of compiled lens modules (opt-in: BLINKY_HIP_CACHE=<directory>).  hiprtc takes
// 0.2-1.1 s per lens, which the engine would feel as a hitch on every first `f_lens`; a code object is keyed by
// FNV-1a-64 over the generated source, the embedded headers, the target arch and the library version.
static uint64_t fnv1a64(const void *data, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
// Where compiled lens modules are kept: bk_set_cache_dir() if the host called it, else $BLINKY_HIP_CACHE ("" or "off"
// disables the disk cache), else $XDG_CACHE_HOME/blinky_hip, else $HOME/.cache/blinky_hip.
static std::mutex g_cache_dir_mutex;
static std::string g_cache_dir;
static bool g_cache_dir_set = false;

extern "C" int bk_set_cache_dir(const char *dir)
{
    std::lock_guard<std::mutex> lock(g_cache_dir_mutex);
    g_cache_dir = dir ? dir : "";
    g_cache_dir_set = dir != nullptr;              // NULL: back to the environment / default location
    return BK_OK;
}

static std::string cache_dir()
{
    {
        std::lock_guard<std::mutex> lock(g_cache_dir_mutex);
        if (g_cache_dir_set) return g_cache_dir;
    }
    if (const char *e = getenv("BLINKY_HIP_CACHE")) return (!*e || !strcmp(e, "off")) ? std::string() : std::string(e);
    if (const char *x = getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/blinky_hip";
    if (const char *h = getenv("HOME")) if (*h) return std::string(h) + "/.cache/blinky_hip";
    return std::string();
}

// ---- what may be loaded from the cache ---------------------------------------------------------------------------------
// A cached object is CODE - GPU code objects, and since round 3 host shared objects that are dlopen()ed into the engine - so the
// cache is only ever a directory that belongs to the user and that nobody else can write to, the files in it likewise, opened
// without following links and checked through the descriptor that is then read / loaded (no check-then-open window).  Names
// carry 128 bits of a SHA-256 over everything the object was built from (source, embedded headers, target / compiler + its
// size and modification time, flags, library version): no collisions to speak of, no stale objects after an upgrade.  Every
// file ends in a trailer - "BKSHA256" + the SHA-256 of what precedes it - that is verified before the object is used: a torn
// or corrupted file is a cache miss, not wrong lensmap entries.  (The digest is not a signature: against somebody who can
// write to the directory only the ownership checks help, which is why a directory that fails them is not used at all.)
namespace {
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    unsigned char buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char *p)
    {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void *data, size_t n)
    {
        const unsigned char *p = (const unsigned char *)data;
        len += n;
        while (n) {
            const size_t take = std::min(n, sizeof buf - fill);
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void update(const std::string &t) { const uint64_t n = t.size(); update(&n, sizeof n); update(t.data(), t.size()); }      // (length-prefixed: "ab","c" != "a","bc")
    void finish(unsigned char out[32])
    {
        const uint64_t bits = len * 8;
        const unsigned char one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        unsigned char l[8];
        for (int i = 0; i < 8; ++i) l[i] = (unsigned char)(bits >> (56 - 8 * i));
        update(l, 8);
        for (int i = 0; i < 8; ++i) { out[4 * i] = (unsigned char)(h[i] >> 24); out[4 * i + 1] = (unsigned char)(h[i] >> 16); out[4 * i + 2] = (unsigned char)(h[i] >> 8); out[4 * i + 3] = (unsigned char)h[i]; }
    }
};
static std::string hex128(Sha256 &s)
{
    unsigned char d[32];
    s.finish(d);
    char out[33];
    for (int i = 0; i < 16; ++i) snprintf(out + 2 * i, 3, "%02x", d[i]);
    return out;
}
static const char kTrailerMagic[8] = {'B', 'K', 'S', 'H', 'A', '2', '5', '6'};
static void trailer_of(const void *data, size_t n, unsigned char out[40])
{
    Sha256 s;
    s.update(data, n);
    memcpy(out, kTrailerMagic, 8);
    s.finish(out + 8);
}
}  // namespace

// a directory this user owns, that nobody else can write to and that is no link; made (0700) if it does not exist
static bool cache_dir_ok(const std::string &dir, bool create)
{
    if (dir.empty()) return false;
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) {
        if (!create) return false;
        const size_t cut = dir.rfind('/');
        if (cut != std::string::npos && cut > 0)
            for (size_t i = 1; i <= cut; ++i)
                if (i == cut || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);        // (the way there: ordinary directories)
        if (mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return false;
        if (lstat(dir.c_str(), &st) != 0) return false;
    }
    return S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// a regular file of this user's that nobody else can write to, opened without following a link; -1 if it is not that
static int open_checked(const std::string &path)
{
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) { close(fd); return -1; }
    return fd;
}
// the whole file behind `fd` minus a valid trailer; false if it is short, torn or altered
static bool read_verified(int fd, std::vector<char> *out)
{
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 40) return false;
    out->resize((size_t)st.st_size);
    size_t got = 0;
    while (got < out->size()) {
        const ssize_t n = pread(fd, out->data() + got, out->size() - got, (off_t)got);
        if (n <= 0) { if (n < 0 && errno == EINTR) continue; return false; }
        got += (size_t)n;
    }
    unsigned char want[40];
    trailer_of(out->data(), out->size() - 40, want);
    if (memcmp(want, out->data() + out->size() - 40, 40) != 0) return false;
    out->resize(out->size() - 40);
    return true;
}
static bool write_all(int fd, const void *data, size_t n)
{
    const char *p = (const char *)data;
    while (n) {
        const ssize_t w = write(fd, p, n);
        if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}

static std::string cache_path(const std::string &source, const std::string &arch)
{
    const std::string dir = cache_dir();
    if (dir.empty()) return std::string();
    Sha256 h;
    h.update(source);
    for (int i = 0; i < bk::kNumEmbeddedHeaders; ++i) h.update(std::string(bk::kEmbeddedHeaders[i].text));
    h.update(arch);
    h.update(std::string(bk_version()));
    h.update(std::string("-O3 -std=c++17 -ffp-contract=off"));
    return dir + "/bk_lens_" + hex128(h) + ".hsaco";
}
static bool cache_load(const std::string &path, std::vector<char> *code)
{
    if (path.empty() || !cache_dir_ok(path.substr(0, path.rfind('/')), false)) return false;
    const int fd = open_checked(path);
    if (fd < 0) return false;
    const bool ok = read_verified(fd, code);
    close(fd);
    return ok;
}
static void cache_store(const std::string &path, const std::vector<char> &code)
{
    if (path.empty() || !cache_dir_ok(path.substr(0, path.rfind('/')), true)) return;       // a cache that cannot be trusted / written is simply not used
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    unsigned char tr[40];
    trailer_of(code.data(), code.size(), tr);
    const bool ok = write_all(fd, code.data(), code.size()) && write_all(fd, tr, sizeof tr);
    close(fd);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

// ---- code objects: memory cache -> disk cache -> hiprtc ------------------------------------------------------------
static std::mutex g_rtc_mutex;                                   // hiprtc + the memory cache
static std::map<uint64_t, std::shared_ptr<std::vector<char>>> g_rtc_cache;

static uint64_t code_key(const std::string &source, const std::string &arch)
{
    return fnv1a64(arch.data(), arch.size(), fnv1a64(source.data(), source.size()));
}

// memory / disk lookups only (cheap); empty result when the source still has to be compiled
static CodeResult cached_code(const std::string &source, const std::string &arch)
{
    CodeResult r;
    std::lock_guard<std::mutex> lock(g_rtc_mutex);
    const bool use_mem = !bk::g_debug.no_memcache;                  // (tests of the disk cache switch the memory cache off)
    auto hit = g_rtc_cache.find(code_key(source, arch));
    if (use_mem && hit != g_rtc_cache.end()) { r.code = hit->second; return r; }
    std::vector<char> code;
    if (cache_load(cache_path(source, arch), &code)) {
        r.code = std::make_shared<std::vector<char>>(std::move(code));
        r.from_disk = true;
        if (g_rtc_cache.size() >= 64) g_rtc_cache.clear();          // (a long session cycling through many lenses)
        g_rtc_cache[code_key(source, arch)] = r.code;
    }
    return r;
}

// one compilation per (source, arch) and process: the stripe contexts of a bk_multi build the same lens side by side
static CodeResult compile_code(const std::string &source, const std::string &arch)
{
    CodeResult r = cached_code(source, arch);
    if (r.code) return r;
    std::lock_guard<std::mutex> lock(g_rtc_mutex);
    auto hit = g_rtc_cache.find(code_key(source, arch));
    if (hit != g_rtc_cache.end() && !bk::g_debug.no_memcache) { r.code = hit->second; return r; }   // another thread was faster
    std::vector<const char *> hnames, htexts;
    for (int i = 0; i < bk::kNumEmbeddedHeaders; ++i) {
        hnames.push_back(bk::kEmbeddedHeaders[i].name);
        htexts.push_back(bk::kEmbeddedHeaders[i].text);
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, source.c_str(), "bk_lens_build.hip", (int)hnames.size(), htexts.data(), hnames.data()) != HIPRTC_SUCCESS) {
        r.rc = BK_E_HIP; r.log = "hiprtcCreateProgram failed";
        return r;
    }
    const char *opts[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off"};
    if (hiprtcCompileProgram(prog, 4, opts) != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, 0);
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        r.rc = BK_E_HIP; r.log = "hiprtc failed to compile the lens kernels: " + log;
        return r;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    r.code = std::make_shared<std::vector<char>>(cs);
    hiprtcGetCode(prog, r.code->data());
    hiprtcDestroyProgram(&prog);
    cache_store(cache_path(source, arch), *r.code);
    if (g_rtc_cache.size() >= 64) g_rtc_cache.clear();
    g_rtc_cache[code_key(source, arch)] = r.code;
    return r;
}


// ---- host module: the generated lens code compiled for the HOST ----------------------------------------------------
// The entries a build flags are re-derived on the platform libm.  The script interpreter does that at 4-10 us per entry;
// the very translation unit hiprtc receives, compiled by the system's C++ compiler against the platform libm
// (bk_hostmod_bkm.h in place of bkm.h, bk_hostmod_driver.inc appended) and dlopen'ed, does it in a fraction of a
// microsecond - the same per-entry functions the kernels call, the same IEEE operations in the same order, libm calls
// resolved in this process's libm.  The compiler runs on another thread and its output is cached next to the device code
// objects; until it is there (or if the machine has no compiler) the interpreter answers, so nothing ever waits for it and
// results are the same either way (tests/test_hostmod.py holds the two against each other entry for entry).
struct HostModule {
    void *dl = nullptr;
    int fd = -1;                       // the descriptor the object was verified and loaded through; kept open: see load_host_module
    void (*inverse)(const BkBuildParams *, const unsigned int *, int, unsigned long, unsigned int *, unsigned char *, int *, int *) = nullptr;
    void (*corners)(const BkBuildParams *, const unsigned int *, int, unsigned long, int *, int *, unsigned char *, int *) = nullptr;
    void (*texel_owns)(const BkBuildParams *, const unsigned int *, int, unsigned long, unsigned char *) = nullptr;
    int (*inverse_scan)(const BkBuildParams *, unsigned int *, unsigned char *, int *) = nullptr;
    ~HostModule() { if (dl) dlclose(dl); if (fd >= 0) close(fd); }
};
using HostModuleP = std::shared_ptr<HostModule>;

static std::mutex g_hostmod_mutex;
static std::map<uint64_t, HostModuleP> g_hostmods;                          // ready (nullptr = failed: do not try again)
static std::map<uint64_t, std::shared_future<HostModuleP>> g_hostmod_jobs;  // compiling
static int g_hostmod_enabled = 1;                                           // bk_set_host_compile

extern "C" int bk_set_host_compile(int on)
{
    std::lock_guard<std::mutex> lock(g_hostmod_mutex);
    g_hostmod_enabled = on != 0;
    return BK_OK;
}

// the C++ compiler to run: $BLINKY_HIP_HOSTCXX ("off" / "" = none), else the first of c++ / g++ / clang++ on $PATH, else ROCm's clang
static std::string host_compiler()
{
    auto on_path = [](const char *name) -> std::string {
        const char *path = getenv("PATH");
        if (!path) return std::string();
        std::string p(path);
        for (size_t b = 0; b <= p.size();) {
            size_t e = p.find(':', b);
            if (e == std::string::npos) e = p.size();
            const std::string cand = p.substr(b, e - b) + "/" + name;
            if (e > b && access(cand.c_str(), X_OK) == 0) return cand;
            b = e + 1;
        }
        return std::string();
    };
    if (const char *e = getenv("BLINKY_HIP_HOSTCXX")) {
        if (!*e || !strcmp(e, "off")) return std::string();
        if (strchr(e, '/')) return access(e, X_OK) == 0 ? std::string(e) : std::string();
        return on_path(e);
    }
    for (const char *name : {"c++", "g++", "clang++"}) { std::string c = on_path(name); if (!c.empty()) return c; }
    for (const char *abs : {"/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/bin/amdclang++"}) if (access(abs, X_OK) == 0) return abs;
    return std::string();
}

static const char *embedded_text(const char *name)
{
    for (int i = 0; i < bk::kNumEmbeddedHeaders; ++i) if (!strcmp(bk::kEmbeddedHeaders[i].name, name)) return bk::kEmbeddedHeaders[i].text;
    return "";
}

// the shared object is loaded THROUGH the descriptor it was checked and verified on (/proc/self/fd): what is mapped is what was read
static HostModuleP load_host_module(const std::string &so)
{
    const int fd = open_checked(so);
    if (fd < 0) return nullptr;
    std::vector<char> content;
    if (!read_verified(fd, &content)) { close(fd); return nullptr; }
    // (the descriptor stays open as long as the module lives: the dynamic loader knows an object by the NAME it was opened under, and a
    //  descriptor number that was closed and handed out again would make "/proc/self/fd/N" answer with the module loaded before)
    const std::string via = "/proc/self/fd/" + std::to_string(fd);
    void *dl = dlopen(via.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) { close(fd); return nullptr; }
    HostModuleP m = std::make_shared<HostModule>();
    m->dl = dl;
    m->fd = fd;
    auto abi = (int (*)(void))dlsym(dl, "bk_hostmod_abi");
    m->inverse = (decltype(m->inverse))dlsym(dl, "bk_hostmod_inverse");
    m->corners = (decltype(m->corners))dlsym(dl, "bk_hostmod_corners");
    m->texel_owns = (decltype(m->texel_owns))dlsym(dl, "bk_hostmod_texel_owns");
    m->inverse_scan = (decltype(m->inverse_scan))dlsym(dl, "bk_hostmod_inverse_scan");
    if (!abi || abi() != 4 + (int)sizeof(BkBuildParams) * 16 || !m->inverse || !m->corners || !m->texel_owns || !m->inverse_scan) return nullptr;
    return m;
}

static bool write_text(const std::string &path, const std::string &text)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
    fclose(f);
    return ok;
}

// compile `source` for the host with `cxx`; the shared object ends up at `so_path` (atomically)
static HostModuleP compile_host_module(const std::string &source, const std::string &cxx, const std::string &so_path)
{
    // a scratch directory nobody else can have prepared: mkdtemp (0700, fails if the name exists) inside the checked cache directory
    std::string tmpl = so_path.substr(0, so_path.rfind('/')) + "/build.XXXXXX";
    if (!mkdtemp(&tmpl[0])) return nullptr;
    const std::string dir = tmpl;
    bool ok = write_text(dir + "/bkm.h", embedded_text("bk_hostmod_bkm.h"));
    for (const char *h : {"bk_build_params.h", "bk_device_rt.h", "bk_build_kernels.h"}) ok = ok && write_text(dir + "/" + h, embedded_text(h));
    const std::string unit = std::string("#define BK_HOST_MODULE 1\n#define __device__\n#define __forceinline__ inline\n") + source + "\n" +
                             embedded_text("bk_hostmod_driver.inc");
    ok = ok && write_text(dir + "/unit.cpp", unit);
    HostModuleP m;
    if (ok) {
        const std::string out = dir + "/unit.so", inc = "-I" + dir, src = dir + "/unit.cpp";
        const char *argv[] = {cxx.c_str(), "-O2", "-std=c++17", "-ffp-contract=off", "-fno-builtin", "-fno-fast-math", "-fPIC", "-shared", "-w",
                              inc.c_str(), "-o", out.c_str(), src.c_str(), nullptr};
        pid_t pid = 0;
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
        posix_spawn_file_actions_addopen(&fa, 2, "/dev/null", O_WRONLY, 0);
        int status = -1;
        if (posix_spawn(&pid, cxx.c_str(), &fa, nullptr, const_cast<char *const *>(argv), environ) == 0) {
            while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
        }
        posix_spawn_file_actions_destroy(&fa);
        if (status == 0) {
            // seal it: the trailer goes behind the ELF image (the loader maps by program headers: trailing bytes are not looked at)
            std::vector<char> img;
            const int fd = open(out.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC);
            struct stat st;
            bool sealed = false;
            if (fd >= 0 && fstat(fd, &st) == 0 && st.st_size > 0) {
                img.resize((size_t)st.st_size);
                if (pread(fd, img.data(), img.size(), 0) == (ssize_t)img.size()) {
                    unsigned char tr[40];
                    trailer_of(img.data(), img.size(), tr);
                    sealed = lseek(fd, 0, SEEK_END) >= 0 && write_all(fd, tr, sizeof tr) && fchmod(fd, 0600) == 0;
                }
            }
            if (fd >= 0) close(fd);
            if (sealed && rename(out.c_str(), so_path.c_str()) == 0) m = load_host_module(so_path);
        }
    }
    for (const char *f : {"bkm.h", "bk_build_params.h", "bk_device_rt.h", "bk_build_kernels.h", "unit.cpp", "unit.so"}) remove((dir + "/" + f).c_str());
    rmdir(dir.c_str());
    return m;
}

// the host module for this generated source: ready -> returned; otherwise a compile is started (once) and nullptr returned,
// unless `wait`.  Never throws; nullptr simply means "use the interpreter".
static HostModuleP host_module_for(const std::string &source, bool wait)
{
    // (source + everything it is compiled with; the headers' part once - they are 100 KB, and this runs in every build that flagged a pixel)
    static const uint64_t headers_key = [] {
        uint64_t k = 1469598103934665603ull;
        for (const char *h : {"bk_hostmod_bkm.h", "bk_hostmod_driver.inc", "bk_build_params.h", "bk_device_rt.h", "bk_build_kernels.h"}) {
            const char *t = embedded_text(h);
            k = fnv1a64(t, strlen(t), k);
        }
        return k;
    }();
    const uint64_t key = fnv1a64(source.data(), source.size(), headers_key);
    std::shared_future<HostModuleP> job;
    {
        std::lock_guard<std::mutex> lock(g_hostmod_mutex);
        if (!g_hostmod_enabled) return nullptr;
        auto hit = g_hostmods.find(key);
        if (hit != g_hostmods.end()) return hit->second;
        auto running = g_hostmod_jobs.find(key);
        if (running != g_hostmod_jobs.end()) job = running->second;
        else {
            const std::string cxx = host_compiler();
            if (cxx.empty()) { g_hostmods[key] = nullptr; return nullptr; }
            std::string dir = cache_dir();                                  // next to the device code objects, or a private temp dir
            if (dir.empty()) {                                              // (disk cache off: a directory only this process can have made)
                static std::string private_dir;
                if (private_dir.empty()) {
                    char tmpl[] = "/tmp/blinky_hip_hostmod.XXXXXX";
                    if (!mkdtemp(tmpl)) { g_hostmods[key] = nullptr; return nullptr; }
                    private_dir = tmpl;
                }
                dir = private_dir;
            }
            if (!cache_dir_ok(dir, true)) { g_hostmods[key] = nullptr; return nullptr; }       // (not a directory to load code from: the interpreter answers)
            // the name: 128 bits of a SHA-256 over the source, what it is compiled with (headers, compiler - path, size, modification
            // time - and flags) and the library version
            Sha256 h;
            h.update(source);
            for (const char *hn : {"bk_hostmod_bkm.h", "bk_hostmod_driver.inc", "bk_build_params.h", "bk_device_rt.h", "bk_build_kernels.h"}) h.update(std::string(embedded_text(hn)));
            h.update(cxx);
            { struct stat cst; if (stat(cxx.c_str(), &cst) == 0) { const long long id[2] = {(long long)cst.st_size, (long long)cst.st_mtime}; h.update(id, sizeof id); } }
            h.update(std::string("-O2 -std=c++17 -ffp-contract=off -fno-builtin -fno-fast-math -fPIC -shared"));
            h.update(std::string(bk_version()));
            const std::string so = dir + "/bk_host_" + hex128(h) + ".so";
            {
                HostModuleP m = load_host_module(so);
                if (m) { g_hostmods[key] = m; return m; }
            }
            job = std::async(std::launch::async, [source, cxx, so]() { return compile_host_module(source, cxx, so); }).share();
            g_hostmod_jobs[key] = job;
        }
    }
    if (!wait && job.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return nullptr;
    HostModuleP m = job.get();
    std::lock_guard<std::mutex> lock(g_hostmod_mutex);
    g_hostmods[key] = m;
    g_hostmod_jobs.erase(key);
    return m;
}

static std::string target_arch(bk_ctx *ctx)
{
    hipDeviceProp_t prop;
    if (ctx->device >= 0 && hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.gcnArchName[0])
        return std::string("--offload-arch=") + prop.gcnArchName;
    return "--offload-arch=gfx950";
}

static int load_module(bk_ctx *ctx, LensProgram *P, const std::string &source, const CodeResult &cr)
{
    if (cr.rc != BK_OK) return ctx->fail(cr.rc, "%s", cr.log.c_str());
    if (P->module) { (void)hipModuleUnload(P->module); P->module = nullptr; }
    P->k_inverse = P->k_corners = P->k_quads = P->k_resolve = P->k_tiles = nullptr;
    P->module_from_cache = cr.from_disk;
    if (ctx->device < 0) {              // host-only context: compiling is all we can do
        P->module_source = source;
        return BK_OK;
    }
    BK_HIP(ctx, hipModuleLoadData(&P->module, cr.code->data()));
    (void)hipModuleGetFunction(&P->k_inverse, P->module, "bk_build_inverse");
    (void)hipModuleGetFunction(&P->k_corners, P->module, "bk_forward_corners");
    (void)hipModuleGetFunction(&P->k_quads, P->module, "bk_forward_quads");
    (void)hipModuleGetFunction(&P->k_resolve, P->module, "bk_forward_resolve");
    if (hipModuleGetFunction(&P->k_tiles, P->module, "bk_forward_tiles") != hipSuccess) P->k_tiles = nullptr;   // (absent under a globe_plate script)
    (void)hipGetLastError();
    P->module_source = source;
    return BK_OK;
}

static int compile_module(bk_ctx *ctx, LensProgram *P, const std::string &source)
{
    if (P->module_source == source && (P->module || ctx->device < 0)) return BK_OK;
    return load_module(ctx, P, source, compile_code(source, target_arch(ctx)));
}

// Compilations nobody waits for any more (the lens or zoom changed while hiprtc was still busy, or the context went away).
// A std::async future blocks in its destructor until its thread is done, so a superseded one is not dropped - that would
// stall the render thread for the rest of the compile - but parked here and reaped once ready; its result still lands in the
// memory / disk caches.  (At process exit the list's destructor joins whatever is still running: defined after the caches
// above, it is destroyed before them.)
static std::mutex g_parked_mutex;
static std::vector<std::shared_future<::CodeResult>> g_parked;
static void park_compile(std::shared_future<::CodeResult> &f)
{
    std::lock_guard<std::mutex> lock(g_parked_mutex);
    for (size_t i = 0; i < g_parked.size();)
        if (g_parked[i].wait_for(std::chrono::seconds(0)) == std::future_status::ready) { g_parked[i] = g_parked.back(); g_parked.pop_back(); }
        else ++i;
    if (f.valid() && f.wait_for(std::chrono::seconds(0)) != std::future_status::ready) g_parked.push_back(f);
    f = std::shared_future<::CodeResult>();
}

// bk_build with asynchronous compilation (bk_set_async_compile): BK_PENDING while hiprtc works on another thread
static int compile_module_async(bk_ctx *ctx, LensProgram *P, const std::string &source)
{
    if (P->module_source == source && P->module) return BK_OK;
    const std::string arch = target_arch(ctx);
    if (P->pending.valid() && P->pending_source == source) {
        if (P->pending.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return BK_PENDING;
        CodeResult cr = P->pending.get();
        P->pending_source.clear();
        P->pending = std::shared_future<::CodeResult>();
        return load_module(ctx, P, source, cr);
    }
    CodeResult cr = cached_code(source, arch);                      // memory / disk: no reason to wait a frame
    if (cr.code) return load_module(ctx, P, source, cr);
    park_compile(P->pending);                                       // (an older source's compile finishes on its own thread)
    P->pending_source = source;
    P->pending = std::async(std::launch::async, [source, arch]() { return compile_code(source, arch); }).share();
    return BK_PENDING;
}

// the asynchronous-compile gate of bk_build / bk_multi_build: is the module for the current lens + globe there?
int bk::build_module_ready(bk_ctx *ctx)
{
    LensProgram *P = ctx->prog;
    if (!(ctx->async_compile && P && P->lens_valid && ctx->globe_valid && P->info.map_type != BK_MAP_NONE)) return BK_OK;
    // a lens that still has to go through hiprtc (0.2-1.1 s): compile on another thread and leave the previous
    // lensmap untouched until the module is there - the caller keeps drawing with it and calls bk_build again
    std::string psrc;
    bk::EmitRequest rq;
    rq.interp = &P->interp; rq.lens_inverse = P->lens_inverse; rq.lens_forward = P->lens_forward; rq.globe_plate = P->globe_plate;
    if (P->emitted.valid && P->emitted.activity == P->interp.activity && P->emitted.libm_rel == bk::g_debug.libm_rel_log2) {
        if (!P->emitted.refused.empty()) return BK_OK;                                       // (a host-path lens: nothing to compile)
        psrc = P->emitted.src;                                                                // (generate_source's answer still stands)
    } else
    try { psrc = bk::emit_build_source(rq); } catch (const LuaError &) { return BK_OK; }      // (reported by the normal path)
    // A module that is ready gets LOADED here (hipModuleLoadData / hipModuleGetFunction): that has to happen on the context's own
    // device, whatever device the calling thread was left on - bk_multi_build asks on behalf of stripe 0 right after a
    // bk_multi_apply whose loop ended on the LAST stripe's device, and kernels loaded there would later be launched on stripe 0's
    // stream (invalid device function on a node with more than one GPU).  The caller's device is put back.
    int prev = -1;
    const bool switched = ctx->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != ctx->device && hipSetDevice(ctx->device) == hipSuccess;
    const int rc = compile_module_async(ctx, P, psrc) == BK_PENDING ? BK_PENDING : BK_OK;
    if (switched) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int bk_set_async_compile(bk_ctx *ctx, int on)
{
    if (!ctx) return BK_E_INVALID;
    ctx->async_compile = on != 0;
    return BK_OK;
}

/* `refused` (nullable): a script whose callbacks use a construct the emitter does not turn into GPU code (recursion, tables made at
 * run time, functions as values, strings ...) is not an error for a caller that passes it - the text of the refusal comes back in
 * it, *out stays empty, and bk_build evaluates the callbacks with the host interpreter instead (build_on_host). */
static int generate_source(bk_ctx *ctx, LensProgram *P, std::string *out, std::string *refused = nullptr)
{
    bk::EmitRequest rq;
    rq.interp = &P->interp;
    rq.lens_inverse = P->lens_inverse;
    rq.lens_forward = P->lens_forward;
    rq.globe_plate = P->globe_plate;
    if (refused) refused->clear();
    if (P->emitted.valid && P->emitted.activity == P->interp.activity && P->emitted.libm_rel == bk::g_debug.libm_rel_log2 &&
        (refused || P->emitted.refused.empty())) {
        *out = P->emitted.src;
        if (refused) *refused = P->emitted.refused;
        return BK_OK;
    }
    P->emitted.valid = false;
    const auto remember = [&](const std::string &src, const std::string &why) {
        P->emitted.src = src; P->emitted.refused = why;
        P->emitted.activity = P->interp.activity; P->emitted.libm_rel = bk::g_debug.libm_rel_log2;
        // callbacks that change nothing a script can see: bk_build may run them (calc_zoom, the flagged entries) and keep this answer
        try { std::string which; P->emitted.stateless = !bk::callbacks_carry_state(rq, &which); } catch (const LuaError &) { P->emitted.stateless = false; }
        P->emitted.valid = true;
    };
    try {
        *out = bk::emit_build_source(rq);
        // test hook: a wider assumed libm discrepancy (tests pair it with bk_set_host_math(ctx, n): the host interpreter on a
        // stand-in libm 2^-n away from bkm.h), so that the flag -> host fix-up path is exercised on thousands of pixels
        if (const int n = bk::g_debug.libm_rel_log2)
            if (n >= 8 && n <= 52) *out = "#define BK_LIBM_REL 0x1p-" + std::to_string(n) + "\n" + *out;
    } catch (const LuaError &e) {
        if (refused && strstr(e.what(), "GPU callback")) {          // (bk_emit.cpp's `unsupported`, and its two "assigned inside a GPU callback but holds a ..." refusals)
            out->clear();
            *refused = e.what();
            remember(*out, *refused);
            return BK_OK;
        }
        return ctx->fail(BK_E_SCRIPT, "%s", e.what());
    }
    P->last_source = *out;
    remember(*out, std::string());
    return BK_OK;
}

#if BK_DEBUG_API
// debug / test hook: the generated HIP translation unit for the current lens + globe
extern "C" int bk_debug_kernel_source(bk_ctx *ctx, char *buf, size_t cap, size_t *needed, int compile)
{
    if (!ctx) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "no valid lens");
    std::string src;
    if (int r = generate_source(ctx, P, &src)) return r;
    if (needed) *needed = src.size() + 1;
    if (buf && cap) snprintf(buf, cap, "%s", src.c_str());
    if (compile) return compile_module(ctx, P, src);
    return BK_OK;
}
#endif

#if BK_DEBUG_API
// debug / test hook: evaluate a callback with the HOST interpreter (the same AST the GPU code was
// generated from).  which: 0 = lens_inverse(x,y), 1 = lens_forward(x,y,z), 2 = globe_plate(x,y,z)
extern "C" int bk_debug_eval(bk_ctx *ctx, int which, const double *args, int nargs, double *out, int *nout)
{
    if (!ctx || !ctx->prog) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    const Value &f = which == 0 ? P->lens_inverse : which == 1 ? P->lens_forward : P->globe_plate;
    if (!f.is_function()) return ctx->fail(BK_E_STATE, "callback not defined");
    try {
        Values a;
        for (int i = 0; i < nargs; ++i) a.push_back(Value::number(args[i]));
        Values r = P->interp.call(f, a);
        *nout = (int)r.size();
        for (size_t i = 0; i < r.size() && i < 8; ++i) out[i] = r[i].t == Value::NUM ? r[i].n : __builtin_nan("");
        if (r.size() == 1 && r[0].t == Value::NIL) *nout = -1;       // a single nil
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "%s", e.what());
    }
    return BK_OK;
}
#endif

#if BK_DEBUG_API
/* test hook: 1 if the current lens module came from the BLINKY_HIP_CACHE directory instead of hiprtc */
extern "C" int bk_debug_module_from_cache(const bk_ctx *ctx) { return ctx && ctx->prog && ctx->prog->module_from_cache ? 1 : 0; }
#endif

extern "C" int bk_set_host_math(bk_ctx *ctx, int portable)
{
    if (!ctx) return BK_E_INVALID;
#if BK_DEBUG_API
    if (portable >= 2) {            // the test suite's stand-in libms (include/blinky_hip_debug.h)
        bk::prog_of(ctx)->interp.math = &math_perturbed(ldexp(1.0, -(portable & 63)), portable >> 6);
        return BK_OK;
    }
#else
    if (portable >= 2) return ctx->fail(BK_E_INVALID, "bk_set_host_math: %d is a test mode of debug-API builds", portable);
#endif
    bk::prog_of(ctx)->interp.math = portable ? &math_portable() : &math_platform();
    return BK_OK;
}

extern "C" const char *bk_script_console(bk_ctx *ctx) { return ctx && ctx->prog ? ctx->prog->console.c_str() : ""; }

// ---- build -----------------------------------------------------------------------------------------------

static void fill_params(bk_ctx *ctx, BkBuildParams *bp)
{
    memset(bp, 0, sizeof *bp);
    bp->W = ctx->W; bp->H = ctx->H;
    bp->row0 = ctx->row0; bp->rows = ctx->rows();
    bp->ps = ctx->ps; bp->gp = ctx->gp; bp->ph = ctx->ph;
    bp->numplates = ctx->numplates;
    bp->has_globe_plate = ctx->prog->globe_plate.is_function();
    bp->scale = ctx->scale;
    bp->inv_scale_up = 1.0 / __builtin_fabs(ctx->scale) * (1.0 + 0x1p-50);
    // set_lensmap_grid constants, fisheye.c:1938-1948 (same double operations)
    const double block_size = ctx->rubix.pad + ctx->rubix.cell;
    const double num_units = ctx->rubix.numcells * block_size + ctx->rubix.pad;
    bp->rubix_block = block_size;
    bp->rubix_pad = ctx->rubix.pad;
    bp->rubix_unit_px = (double)ctx->ps / num_units;
    if (ctx->ps <= 32 * (int)(sizeof bp->grid_bits / sizeof bp->grid_bits[0])) {
        const double key[4] = {(double)ctx->ps, (double)ctx->rubix.numcells, ctx->rubix.cell, ctx->rubix.pad};
        if (memcmp(key, ctx->grid_cache_key, sizeof key) != 0) {
            memset(ctx->grid_cache, 0, sizeof ctx->grid_cache);
            for (int p = 0; p < ctx->ps; ++p)            /* fisheye.c:1950-1957; division and fmod are exact operations */
                if (__builtin_fmod((double)p / bp->rubix_unit_px, bp->rubix_block) < bp->rubix_pad) ctx->grid_cache[p >> 5] |= 1u << (p & 31);
            memcpy(ctx->grid_cache_key, key, sizeof key);
        }
        static_assert(sizeof ctx->grid_cache == sizeof bp->grid_bits, "one bitmap");
        memcpy(bp->grid_bits, ctx->grid_cache, sizeof bp->grid_bits);
        bp->grid_n = ctx->ps;
    }
    for (int i = 0; i < ctx->numplates; ++i) {
        const bk_plate &p = ctx->plates[i];
        BkPlateDev &d = bp->plates[i];
        memcpy(d.forward, p.forward, 12); memcpy(d.right, p.right, 12); memcpy(d.up, p.up, 12);
        d.fov = p.fov; d.dist = p.dist;
        d.dist64 = 0.5 / ctx->prog->interp.math->tan((double)(p.fov / 2));      /* fisheye.c:2060 */
    }
    bp->offsets = ctx->d_offsets;
    bp->tints = ctx->d_tints;
    bp->display = ctx->d_display;
    bp->err = ctx->d_display + BK_MAX_PLATES;
    bp->flag_count = (unsigned int *)(ctx->d_display + BK_MAX_PLATES + 1);
    bp->first_bad = (unsigned int *)(ctx->d_display + BK_MAX_PLATES + 2);
    bp->flag_list = ctx->d_flag_list;
    bp->flag_cap = (unsigned int)ctx->flag_cap;
}

static const char *err_text(int bits)
{
    if (bits & BK_ERR_LOOP) return "a lens callback exceeded the per-pixel iteration budget (infinite loop?)";
    if (bits & BK_ERR_ARITH) return "a lens callback performed arithmetic on a non-number (nil?)";
    if (bits & BK_ERR_COMPARE) return "a lens callback compared non-numbers with < or <=";
    if (bits & BK_ERR_INDEX) return "a lens callback stored outside a table's bounds";
    if (bits & BK_ERR_RESULT) return "a lens callback returned a malformed result (not 3 numbers / 2 numbers / a single nil)";
    return "unknown device error";
}

// ---- host re-evaluation of the entries a build flagged ------------------------------------------------------
// The device evaluates the callbacks on bkm.h; the reference's Lua VM calls the platform libm.  The kernels
// flag every pixel / texel corner / texel whose DISCRETE outcome could depend on the last bits of a
// transcendental (bk_device_rt.h) - a handful per 4K map - and the functions below re-derive exactly those
// with this context's host interpreter (platform libm unless bk_set_host_math says otherwise: the same
// evaluator calc_zoom and the globe loader use), restating fisheye.c's float/double tail, and patch the table.
namespace {

float h_dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }     /* mathlib.h:70 */
/* who evaluates: the context's interpreter, or a per-thread copy of it (Interp::clone) */
struct HostEval {
    Interp *I;
    Value lens_inverse, lens_forward, globe_plate;
    Values call(const Value &f, const Values &a) { I->steps = 0; return I->call(f, a); }
};

/* ray_to_plate_index, fisheye.c:2023-2050 (same out-of-range handling as the kernel's) */
int h_ray_to_plate_index(HostEval &E, const BkBuildParams &bp, const float *ray)
{
    if (E.globe_plate.is_function()) {
        Values r = E.call(E.globe_plate, Values{Value::number((double)ray[0]), Value::number((double)ray[1]),
                                                Value::number((double)ray[2])});
        if (r.empty() || r.back().t != Value::NUM) return -1;
        const double d = r.back().n;
        if (!(d > -2147483648.0 && d < 2147483647.0)) return -1;
        const int plate = (int)__builtin_rint(d);
        if (plate < 0 || plate >= bp.numplates) return -1;
        return plate;
    }
    int plate_index = 0;
    double max_dp = -2;
    for (int i = 0; i < bp.numplates; ++i) {
        const double dp = (double)h_dot3(ray, bp.plates[i].forward);
        if (dp > max_dp) { max_dp = dp; plate_index = i; }
    }
    return plate_index;
}

bool h_offgrid(const BkBuildParams &bp, int px, int py)                                                /* fisheye.c:1922-1960 */
{
    const double ux = (double)px / bp.rubix_unit_px, uy = (double)py / bp.rubix_unit_px;
    return !(__builtin_fmod(ux, bp.rubix_block) < bp.rubix_pad || __builtin_fmod(uy, bp.rubix_block) < bp.rubix_pad);
}

/* one pixel of resume_lensmap_inverse (fisheye.c:2084-2124, 1545-1588, 1995-2013); o = index inside the owned stripe */
void h_inverse_entry(HostEval &E, const BkBuildParams &bp, uint32_t o, uint32_t *off, uint8_t *tint,
                     int *plate_shown, int *err)
{
    *off = BK_NULL_OFFSET; *tint = 255; *plate_shown = -1;
    const int lyl = (int)(o / (uint32_t)bp.W), lx = (int)(o - (uint32_t)lyl * (uint32_t)bp.W), ly = bp.row0 + lyl;
    const double y = (double)(-(ly - bp.H / 2)) * bp.scale, x = (double)(lx - bp.W / 2) * bp.scale;
    Values r = E.call(E.lens_inverse, Values{Value::number(x), Value::number(y)});
    if (r.size() == 3 && r[0].t == Value::NUM && r[1].t == Value::NUM && r[2].t == Value::NUM) {
        float ray[3] = {(float)r[0].n, (float)r[1].n, (float)r[2].n};
        bk::h_vector_normalize(ray);
        const int plate = h_ray_to_plate_index(E, bp, ray);
        if (plate < 0) return;
        const BkPlateDev &p = bp.plates[plate];
        const double px_ = (double)h_dot3(p.right, ray), py_ = (double)h_dot3(p.up, ray), pz_ = (double)h_dot3(p.forward, ray);
        const double u = px_ / pz_ * p.dist64 + 0.5, v = -py_ / pz_ * p.dist64 + 0.5;
        if (u >= 0 && u <= 1 && v >= 0 && v <= 1) {
            const int px = h_trunc_to_int(u * bp.ps), py = h_trunc_to_int(v * bp.ps);
            if (px >= 0 && px < bp.ps && py >= 0 && py < bp.ps) {
                *plate_shown = plate;
                *off = bk_texel_offset((unsigned)bp.gp, (unsigned)bp.ph, (unsigned)plate, (unsigned)px, (unsigned)py);
                if (h_offgrid(bp, px, py)) *tint = (uint8_t)plate;
            }
        }
    } else if (!(r.size() == 1 && r[0].t == Value::NIL)) {
        *err |= BK_ERR_RESULT;
    }
}

/* one texel corner of the forward build: uv_to_screen, fisheye.c:2227-2243 */
void h_corner_entry(bk_ctx *ctx, HostEval &E, const BkBuildParams &bp, uint32_t id, int *sx, int *sy, uint8_t *ok, int *err)
{
    const uint32_t n1 = (uint32_t)bp.ps + 1u;
    const uint32_t plate = id / (n1 * n1), rem = id - plate * n1 * n1, j = rem / n1, i = rem - j * n1;
    const double u = ((double)i - 0.5) / bp.ps, v = ((double)j - 0.5) / bp.ps;
    float ray[3];
    bk::h_plate_uv_to_ray(ctx->plates[plate], u, v, ray);
    *sx = 0; *sy = 0; *ok = 0;
    Values r = E.call(E.lens_forward, Values{Value::number((double)ray[0]), Value::number((double)ray[1]),
                                              Value::number((double)ray[2])});
    if (r.size() == 2 && r[0].t == Value::NUM && r[1].t == Value::NUM) {
        *sx = h_trunc_to_int(r[0].n / bp.scale + (double)(bp.W / 2));
        *sy = h_trunc_to_int(-r[1].n / bp.scale + (double)(bp.H / 2));
        *ok = 1;
    } else if (!(r.size() == 1 && r[0].t == Value::NIL)) {
        *err |= BK_ERR_RESULT;
    }
}

/* forward build: does the ray through texel `id` select its own plate (fisheye.c:2193-2196) */
bool h_texel_owns(bk_ctx *ctx, HostEval &E, const BkBuildParams &bp, uint32_t id)
{
    const uint32_t ps = (uint32_t)bp.ps, plate = id / (ps * ps), rem = id - plate * ps * ps, py = rem / ps, px = rem - py * ps;
    float ray[3];
    bk::h_plate_uv_to_ray(ctx->plates[plate], (double)px / bp.ps, (double)py / bp.ps, ray);
    return (int)plate == h_ray_to_plate_index(E, bp, ray);
}

/* A small process-wide pool of worker threads for the host re-evaluation (creating 64-256 threads per build cost more
 * than the evaluations themselves).  Created on first use; its threads sleep on a condition variable and are JOINED when
 * the library is unloaded or the process exits (a function-local static: destroyed by dlclose / exit handlers, after which
 * no code of this library runs any more). */
class FixupPool {
public:
    static FixupPool &get()
    {
        static FixupPool pool;
        return pool;
    }
    size_t size() const { return threads_.size(); }
    /* run job(worker_index) on UP TO `want` workers (<= size() + 1; the caller is one of them) and wait for those that started: every job
     * passed here draws its work from a counter its workers share and returns when that is used up - a worker index that never gets its
     * turn is no loss */
    void run(size_t want, const std::function<void(size_t)> &job)
    {
        if (!want) return;
        std::unique_lock<std::mutex> submit(submit_mutex_);  // one parallel section at a time
        // The caller is the last worker: the section starts now, not when the first sleeper has woken up (waking forty sleepers through
        // one mutex takes longer than a short section's whole work - the jobs here all draw their work from a shared counter, so whoever
        // arrives late simply takes less).
        const size_t pooled = want - 1;
        if (pooled) {
            {
                std::lock_guard<std::mutex> lock(m_);
                job_ = &job;
                want_ = pooled;
                next_ = 0;
                done_ = 0;
                ++generation_;
            }
            if (pooled >= threads_.size() / 2) wake_.notify_all();
            else for (size_t i = 0; i < pooled; ++i) wake_.notify_one();
        }
        job(pooled);
        if (pooled) {
            // (the caller's job returns when the shared counter is used up: workers that have not even woken yet are not waited for -
            //  their turns are withdrawn - only those that are in the middle of their last piece)
            std::unique_lock<std::mutex> lock(m_);
            want_ = next_;
            finished_.wait(lock, [&] { return done_ == want_; });
            job_ = nullptr;
            want_ = next_ = done_ = 0;
        }
    }
    ~FixupPool()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread &t : threads_) if (t.joinable()) t.join();
    }

private:
    FixupPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        size_t workers = std::min<size_t>(hw ? hw : 1, 128);
        if (const char *e = getenv("BLINKY_HIP_FIXUP_THREADS")) workers = (size_t)std::max(1, std::min(256, atoi(e)));
        for (size_t i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
    }
    void loop()
    {
        for (;;) {
            const std::function<void(size_t)> *job = nullptr;
            size_t mine = 0;
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return stop_ || next_ < want_; });
                if (stop_) return;
                job = job_;
                mine = next_++;
            }
            (*job)(mine);
            {
                std::lock_guard<std::mutex> lock(m_);
                if (++done_ == want_) finished_.notify_all();
            }
        }
    }
    std::mutex m_, submit_mutex_;
    std::condition_variable wake_, finished_;
    const std::function<void(size_t)> *job_ = nullptr;
    size_t want_ = 0, next_ = 0, done_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
    std::vector<std::thread> threads_;
};

/* the flag list (records of four words, [0] = entry id) in ascending id order: the order the reference's scan meets the
 * entries in (fisheye.c:2084-2124), which is also what keeps a script's own caches warm - eckert4 solves its outline once
 * per ROW and remembers it in globals; met in the arbitrary order the kernel appended them every flagged pixel would solve
 * it again (20 Newton steps), an order of magnitude more host time on the rows it flags whole */
void sort_flagged(std::vector<uint32_t> &rec)
{
    const size_t n = rec.size() / 4;
    if (n < 2) return;
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; ++i) sorted = rec[4 * (i - 1)] <= rec[4 * i];
    if (sorted) return;
    std::vector<uint64_t> key(n), tmp(n);
    for (size_t i = 0; i < n; ++i) key[i] = ((uint64_t)rec[4 * i] << 32) | (uint64_t)i;
    if (n < 4096) std::sort(key.begin(), key.end());
    else {
        for (int shift = 32; shift < 64; shift += 11) {           // LSD radix on the id, 11 bits a pass (stable)
            size_t count[2049] = {0};
            for (size_t i = 0; i < n; ++i) ++count[((key[i] >> shift) & 2047u) + 1];
            for (int b = 0; b < 2048; ++b) count[b + 1] += count[b];
            for (size_t i = 0; i < n; ++i) tmp[count[(key[i] >> shift) & 2047u]++] = key[i];
            key.swap(tmp);
        }
    }
    std::vector<uint32_t> out(rec.size());
    for (size_t i = 0; i < n; ++i) memcpy(&out[4 * i], &rec[4 * (size_t)(key[i] & 0xFFFFFFFFu)], 16);
    rec.swap(out);
}

/* fn(E, i) for i in [0, n): on the context's interpreter when the list is short, otherwise on the pool's workers, each
 * on its own deep copy of the interpreter state; the workers take runs of consecutive entries from a shared counter (the
 * entries cost very different amounts - a nil-test that fails at once, a Newton iteration - so equal static shares leave
 * most workers waiting for the unluckiest).  The first script error any worker meets is rethrown here. */
template <typename Fn>
void for_each_flagged(LensProgram *P, size_t n, Fn fn)
{
    // (never on the context's own interpreter: a callback that assigns script globals - fahey's `lat`, `lon` - would leave them
    //  changed, the next build's generated source would carry different initial values for them, and the lens would go through
    //  hiprtc again for nothing: the 490 ms second build of fahey in round 2's tables)
    if (!n) return;                                 // (nothing flagged - most builds: no copy of the interpreter, 30 us at 4K panini)
    const Values roots_in{P->lens_inverse, P->lens_forward, P->globe_plate};
    FixupPool *pool_p = n >= 256 ? &FixupPool::get() : nullptr;
    const size_t nthreads = pool_p ? std::max<size_t>(1, std::min(pool_p->size(), n / 96)) : 1;
    if (nthreads <= 1) {
        Values roots;
        std::unique_ptr<Interp> mine = P->interp.clone(roots_in, &roots);
        HostEval ev{mine.get(), roots[0], roots[1], roots[2]};
        for (size_t i = 0; i < n; ++i) fn(ev, i);
        return;
    }
    FixupPool &pool = *pool_p;
    const size_t run = std::max<size_t>(32, std::min<size_t>(2048, n / (nthreads * 6)));
    std::atomic<size_t> next{0};
    std::vector<std::string> errors(nthreads);
    pool.run(nthreads, [&](size_t t) {
        try {
            // every worker copies the interpreter for itself (the original is only read meanwhile)
            Values roots;
            std::unique_ptr<Interp> mine = P->interp.clone(roots_in, &roots);
            HostEval ev{mine.get(), roots[0], roots[1], roots[2]};
            for (;;) {
                const size_t i0 = next.fetch_add(run, std::memory_order_relaxed);
                if (i0 >= n) break;
                for (size_t i = i0, e = std::min(n, i0 + run); i < e; ++i) fn(ev, i);
            }
        } catch (const LuaError &e) { errors[t] = e.what(); }
    });
    for (const std::string &e : errors) if (!e.empty()) throw LuaError(e);
}

/* call(first, count) over [0, n) in runs of consecutive entries, on a few pool workers: the compiled host module answers an
 * entry in a fraction of a microsecond, so a handful of threads is plenty */
template <typename Call>
void hostmod_runs(size_t n, Call call)
{
    const size_t run = n < 65536 ? 128 : 512;       // (short lists in short runs: the last one to finish decides)
    FixupPool &pool = FixupPool::get();
    // (two runs per worker at least; r6: the cap was 48 below 200 K entries - waking more sleepers than that cost more than they brought
    //  while every one of them was waited for; measured at 4K: quincuncial's 20 132 entries 0.67-0.94 -> 0.49-0.55 ms, eckert4's 7 707 0.13 -> 0.12,
    //  winkeltripel's 10 431 0.50 -> 0.40-0.58: what is left is the sleepers' wake-up, one after the other)
    const size_t nthreads = n < 2 * run ? 1 : std::min<size_t>(std::min<size_t>(pool.size(), (n + 2 * run - 1) / (2 * run)), 128);
    if (nthreads <= 1) { if (n) call((size_t)0, n); return; }
    std::atomic<size_t> next{0};
    pool.run(nthreads, [&](size_t) {
        for (;;) {
            const size_t i0 = next.fetch_add(run, std::memory_order_relaxed);
            if (i0 >= n) break;
            call(i0, std::min(run, n - i0));
        }
    });
}

}  // namespace

void bk::host_parallel(size_t parts, const std::function<void(size_t)> &job)
{
    if (parts <= 1) { if (parts) job(0); return; }
    FixupPool &pool = FixupPool::get();
    const size_t nthreads = std::min(parts, pool.size());
    if (nthreads <= 1) { for (size_t i = 0; i < parts; ++i) job(i); return; }
    std::atomic<size_t> next{0};
    pool.run(nthreads, [&](size_t) {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= parts) break;
            job(i);
        }
    });
}

/* the compiled host module to re-derive flagged entries with, or nullptr: the interpreter does it (host math switched away
 * from the platform libm, no compiler on this machine, still compiling, switched off) */
static HostModuleP fixup_module(LensProgram *P, const std::string &source)
{
    if (bk::g_debug.host_module == 2) return nullptr;
    if (P->interp.math != &math_platform()) return nullptr;
    return host_module_for(source, bk::g_debug.host_module == 1);
}

// the flag list a kernel just filled: grows the list and reports `retry` when it overflowed
static int read_flagged(bk_ctx *ctx, unsigned int count, std::vector<uint32_t> *list, bool *retry)
{
    *retry = false;
    list->clear();
    if (count > ctx->flag_cap) {
        (void)hipFree(ctx->d_flag_list);
        ctx->d_flag_list = nullptr;
        ctx->flag_cap = 0;
        const size_t cap = (size_t)count + count / 4 + 1024;
        BK_HIP(ctx, hipMalloc((void **)&ctx->d_flag_list, cap * 4 * sizeof(uint32_t)));
        ctx->flag_cap = cap;
        *retry = true;
        return BK_OK;
    }
    if (!count) return BK_OK;
    list->resize((size_t)count * 4);
    BK_HIP(ctx, hipMemcpyAsync(list->data(), ctx->d_flag_list, (size_t)count * 16, hipMemcpyDeviceToHost, ctx->stream));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // The kernels append wave by wave: runs of up to 64 ascending ids.  A short list is put into the reference's scan order outright;
    // a long one (eckert4 flags whole rows: 640 K entries) is left in its runs - sorting it cost more than the re-derivation, and a
    // script's per-row cache is as warm within a run as within the sorted list.
    if (count <= 131072) sort_flagged(*list);
    return BK_OK;
}

/* Where a host-built table goes: into the context's device lensmap - or, for bk_debug_host_build on a context without a device, into the
 * caller's arrays (ctx->host_sink_*).  `off` == nullptr: the empty table. */
static int deliver_host_table(bk_ctx *ctx, const uint32_t *off, const uint8_t *tint)
{
    const size_t px = (size_t)ctx->W * ctx->rows();
    if (ctx->host_sink_off) {
        if (off) { memcpy(ctx->host_sink_off, off, px * 4); memcpy(ctx->host_sink_tint, tint, px); }
        else { memset(ctx->host_sink_off, 0xFF, px * 4); memset(ctx->host_sink_tint, 255, px); }
        return BK_OK;
    }
    if (off) {
        BK_HIP(ctx, hipMemcpyAsync(ctx->d_offsets, off, px * 4, hipMemcpyHostToDevice, ctx->stream));
        BK_HIP(ctx, hipMemcpyAsync(ctx->d_tints, tint, px, hipMemcpyHostToDevice, ctx->stream));
    } else {
        BK_HIP(ctx, hipMemsetAsync(ctx->d_offsets, 0xFF, px * 4, ctx->stream));
        BK_HIP(ctx, hipMemsetAsync(ctx->d_tints, 255, px, ctx->stream));
    }
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BK_OK;
}

// The inverse build as ONE sequential scan on the host (bk_set_sequential_build): what the reference does (fisheye.c:2084-2124) -
// one evaluator whose script globals travel from pixel to pixel, rows from the bottom up, pixels left to right, stopping at the
// first malformed result - by the compiled host module on the platform libm (waited for), else by the script interpreter.  For
// scripts that accumulate state across pixels, which the parallel GPU build cannot reproduce.  A stripe context scans its own rows
// with a fresh state: use a single full-height context for such scripts.
static int build_sequential(bk_ctx *ctx, LensProgram *P, const std::string &source, const BkBuildParams &bp, int display_out[BK_MAX_PLATES])
{
    const size_t px = (size_t)ctx->W * ctx->rows();
    std::vector<uint32_t> off(px, BK_NULL_OFFSET);
    std::vector<uint8_t> tint(px, 255);
    int disp[BK_MAX_PLATES] = {0, 0, 0, 0, 0, 0};
    int errbits = 0;
    const auto t0 = std::chrono::steady_clock::now();
    // (no generated source - the emitter declined the script, build_on_host - means no compiled host module either: the interpreter scans)
    HostModuleP hm = !source.empty() && P->interp.math == &math_platform() && bk::g_debug.host_module != 2 ? host_module_for(source, true) : nullptr;
    ctx->last_fixup_compiled = hm != nullptr;
    try {
        if (hm) errbits = hm->inverse_scan(&bp, off.data(), tint.data(), disp);
        else {
            const Values roots_in{P->lens_inverse, P->lens_forward, P->globe_plate};
            Values roots;
            std::unique_ptr<Interp> mine = P->interp.clone(roots_in, &roots);
            HostEval ev{mine.get(), roots[0], roots[1], roots[2]};
            for (int lyl = ctx->rows() - 1; lyl >= 0 && !errbits; --lyl)
                for (int lx = 0; lx < ctx->W && !errbits; ++lx) {
                    const size_t o = (size_t)lyl * ctx->W + lx;
                    int shown = -1;
                    h_inverse_entry(ev, bp, (uint32_t)o, &off[o], &tint[o], &shown, &errbits);
                    if (errbits) {
                        off[o] = BK_NULL_OFFSET; tint[o] = 255;
                        if (errbits == BK_ERR_RESULT) ctx->last_bad_key = (uint32_t)(((uint32_t)ctx->row0 + (uint32_t)lyl) * (uint32_t)ctx->W + ((uint32_t)ctx->W - 1u - (uint32_t)lx)) + 1u;
                    }
                    else if (shown >= 0) disp[shown] = 1;
                }
        }
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", e.what());
    }
    ctx->last_host_eval_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->last_build_ms = ctx->last_host_eval_ms;
    ctx->last_flagged = ctx->last_changed = 0;
    if (errbits & ~BK_ERR_RESULT) {                              // a Lua runtime error: nothing is drawn (see bk_build)
        std::fill(off.begin(), off.end(), BK_NULL_OFFSET);
        std::fill(tint.begin(), tint.end(), (uint8_t)255);
        for (int &d : disp) d = 0;
    }
    if (int r = deliver_host_table(ctx, off.data(), tint.data())) return r;
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = i < ctx->numplates ? disp[i] : 0; if (display_out) display_out[i] = ctx->display[i]; }
    if (errbits) return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", err_text(errbits));
    return BK_OK;
}

// ---- the HOST build: scripts whose callbacks the emitter declines ---------------------------------------------------------------
// The reference lua_calls whatever the script defines (fisheye.c:1551, 1597, 1640).  The emitter turns most of Lua into GPU code,
// not all of it: recursion, tables made at run time, functions as values, strings.  Such a script is not refused: its callbacks are
// evaluated by the product's own interpreter (the evaluator calc_zoom, the globe loader and the flagged-pixel fix-up use) - on the
// worker pool, every worker on its own copy of the script state, when the callbacks provably carry nothing from one call to the
// next (bk_lens_carries_state = 0), else as ONE scan in the reference's order.  Slower by orders of magnitude than the kernels
// (seconds at 4K), and the reference's result.  Nothing here is a CPU fallback of the GPU work: the table it builds is uploaded and
// applied by the same kernels; what runs on the host is the user's Lua, which only an interpreter can run.

/* set_lensmap_from_plate (fisheye.c:1963-1982) on host tables, stripe-filtered like the kernels' commit; the tint of an on-grid
 * writer leaves an earlier off-grid writer's tint in place (set_lensmap_grid only ever sets it: 1957-1958) */
namespace {
struct HostTable { const BkBuildParams *bp; uint32_t *off; uint8_t *tint; int *disp; };
inline void h_fwd_set(const HostTable &T, int lx, int ly, int px, int py, int plate, bool offgrid)
{
    const BkBuildParams &P = *T.bp;
    if (lx < 0 || lx >= P.W || ly < 0 || ly >= P.H) return;
    T.disp[plate] = 1;                                         /* display is global, not per stripe (bk_fwd_set) */
    if (ly < P.row0 || ly >= P.row0 + P.rows) return;
    const size_t o = (size_t)(ly - P.row0) * P.W + lx;
    T.off[o] = bk_texel_offset((unsigned)P.gp, (unsigned)P.ph, (unsigned)plate, (unsigned)px, (unsigned)py);
    if (offgrid) T.tint[o] = (uint8_t)plate;
}
inline int h_wrap_sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
/* draw_quad (fisheye.c:2246-2338) with the x86-64 build's wrap-around on INT_MIN corners; the loops visit the visible part of a
 * 2^31-long range only - the host twin of bk_draw_quad (bk_build_kernels.h), which says why that is bit-identical */
void h_draw_quad(const HostTable &T, const int *tl, const int *tr, const int *bl, const int *br, int plate, int px, int py, bool offgrid)
{
    const BkBuildParams &P = *T.bp;
    const int *p[4] = {tl, tr, br, bl};
    int x = tl[0], y = tl[1];
    int miny = y, maxy = y, minx = x, maxx = x;
    for (int i = 1; i < 4; i++) {
        const int tx = p[i][0], ty = p[i][1];
        if (tx < minx) minx = tx; else if (tx > maxx) maxx = tx;
        if (ty < miny) miny = ty; else if (ty > maxy) maxy = ty;
    }
    const int maxdiff = 20;
    {
        const int dx = h_wrap_sub(minx, maxx), dy = h_wrap_sub(miny, maxy);
        const int adx = dx < 0 ? h_wrap_sub(0, dx) : dx, ady = dy < 0 ? h_wrap_sub(0, dy) : dy;
        if (adx > maxdiff || ady > maxdiff) return;                                       /* :2272 */
    }
    const int vx0 = minx < 0 ? 0 : minx, vx1 = maxx >= P.W ? P.W - 1 : maxx;
    const int vy0 = miny < 0 ? 0 : miny, vy1 = maxy >= P.H ? P.H - 1 : maxy;
    if (miny == maxy && minx == maxx) { h_fwd_set(T, x, y, px, py, plate, offgrid); return; }
    if (miny == maxy) { for (int tx = vx0; tx <= vx1; ++tx) h_fwd_set(T, tx, miny, px, py, plate, offgrid); return; }
    if (minx == maxx) { for (int ty = vy0; ty <= vy1; ++ty) h_fwd_set(T, x, ty, px, py, plate, offgrid); return; }
    const bool tall = h_wrap_sub(maxy, miny) < 0;
    const int y_first = tall ? vy0 : miny, nrows = h_wrap_sub(tall ? vy1 : maxy, y_first);
    for (int ky = 0; ky <= nrows; ++ky) {
        y = (int)((unsigned)y_first + (unsigned)ky);
        int tx[2] = {minx, maxx};
        int txi = 0, j = 3;
        for (int i = 0; i < 4; ++i) {
            const int ix = p[i][0], iy = p[i][1], jx = p[j][0], jy = p[j][1];
            if ((iy < y && y <= jy) || (jy < y && y <= iy)) {                             /* :2310 */
                const double dy = (double)h_wrap_sub(jy, iy), dx = (double)h_wrap_sub(jx, ix);
                tx[txi] = h_trunc_to_int((double)ix + (double)h_wrap_sub(y, iy) / dy * dx); /* :2313 */
                if (++txi == 2) break;
            }
            j = i;
        }
        if (tx[0] > tx[1]) { const int t = tx[0]; tx[0] = tx[1]; tx[1] = t; }
        if (h_wrap_sub(tx[1], tx[0]) > maxdiff) return;                                   /* :2327 aborts the quad */
        const int x_first = tx[0] < 0 ? 0 : tx[0], x_last = tx[1] >= P.W ? P.W - 1 : tx[1];
        for (x = x_first; x <= x_last; ++x) h_fwd_set(T, x, y, px, py, plate, offgrid);
    }
}
}  // namespace

/* what a host build leaves in the context: the table on the device, display flags, timings, and how it was built */
static int finish_host_build(bk_ctx *ctx, const std::vector<uint32_t> &off, const std::vector<uint8_t> &tint, const int disp[BK_MAX_PLATES],
                             int display_out[BK_MAX_PLATES], std::chrono::steady_clock::time_point t0)
{
    ctx->last_host_eval_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->last_build_ms = ctx->last_host_eval_ms;
    ctx->last_flagged = ctx->last_changed = 0;
    if (int r = deliver_host_table(ctx, off.data(), tint.data())) return r;
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = i < ctx->numplates ? disp[i] : 0; if (display_out) display_out[i] = ctx->display[i]; }
    return BK_OK;
}
static int empty_host_build(bk_ctx *ctx, int display_out[BK_MAX_PLATES], const char *what)
{
    // a Lua run-time error, which the reference's unprotected lua_call does not survive: nothing is drawn (see bk_build)
    if (int r = deliver_host_table(ctx, nullptr, nullptr)) return r;
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = 0; if (display_out) display_out[i] = 0; }
    return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", what);
}

/* resume_lensmap_inverse (fisheye.c:2084-2124) on the worker pool: every pixel by the interpreter, every worker on its own copy of
 * the script state; a malformed result ends the reference's scan there, so what lies behind it in scan order is taken away again
 * (the kernels' rule: launch_truncate_scan) */
static int build_inverse_pool(bk_ctx *ctx, LensProgram *P, const BkBuildParams &bp, int display_out[BK_MAX_PLATES])
{
    const size_t px = (size_t)ctx->W * ctx->rows();
    std::vector<uint32_t> off(px, BK_NULL_OFFSET);
    std::vector<uint8_t> tint(px, 255);
    std::vector<int8_t> shown(px, (int8_t)-1);
    std::vector<uint8_t> err(px, 0);
    int disp[BK_MAX_PLATES] = {0, 0, 0, 0, 0, 0};
    const auto t0 = std::chrono::steady_clock::now();
    try {
        for_each_flagged(P, px, [&](HostEval &E, size_t o) {
            int s = -1, e = 0;
            h_inverse_entry(E, bp, (uint32_t)o, &off[o], &tint[o], &s, &e);
            shown[o] = (int8_t)s; err[o] = (uint8_t)e;
        });
    } catch (const LuaError &e) {
        return empty_host_build(ctx, display_out, e.what());
    }
    uint32_t bad_key = 0;
    for (size_t o = 0; o < px; ++o) {
        if (!err[o]) continue;
        if (err[o] & ~BK_ERR_RESULT) return empty_host_build(ctx, display_out, err_text(err[o]));
        const uint32_t lyl = (uint32_t)(o / (size_t)ctx->W), lx = (uint32_t)(o - (size_t)lyl * ctx->W);
        bad_key = std::max(bad_key, (uint32_t)(((uint32_t)ctx->row0 + lyl) * (uint32_t)ctx->W + ((uint32_t)ctx->W - 1u - lx)) + 1u);
    }
    if (bad_key) {
        // the scan reaches pixel (ly, lx) before the failing one iff its key is larger: everything else - the failing pixel too - is NULL
        for (size_t o = 0; o < px; ++o) {
            const uint32_t lyl = (uint32_t)(o / (size_t)ctx->W), lx = (uint32_t)(o - (size_t)lyl * ctx->W);
            const uint32_t key = (uint32_t)(((uint32_t)ctx->row0 + lyl) * (uint32_t)ctx->W + ((uint32_t)ctx->W - 1u - lx)) + 1u;
            if (key <= bad_key) { off[o] = BK_NULL_OFFSET; tint[o] = 255; shown[o] = -1; }
        }
    }
    for (size_t o = 0; o < px; ++o) if (shown[o] >= 0) disp[shown[o]] = 1;
    ctx->last_bad_key = bad_key;
    if (int r = finish_host_build(ctx, off, tint, disp, display_out, t0)) return r;
    if (bad_key) return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", err_text(BK_ERR_RESULT));
    return BK_OK;
}

/* resume_lensmap_forward (fisheye.c:2126-2217) on the host.  sequential: ONE evaluator and the reference's own call order - per
 * plate the bottom row of corners, then per texel row (from the last up) its upper corners left to right and its texels' "own
 * plate" tests (globe_plate) left to right - so that state a script carries from call to call travels as in the reference.
 * Otherwise a plate's corners (and its own-plate tests, with a globe_plate script) are evaluated on the worker pool and the
 * quads are drawn afterwards in the reference's order (later writers overwrite).  A nil corner skips the quads that touch it
 * (the reference reads a stale entry there; no shipped lens returns nil: DESIGN.md 5); a malformed
 * result leaves an EMPTY lensmap, as the kernels' forward build does (blinky_hip.h). */
static int build_forward_host(bk_ctx *ctx, LensProgram *P, const BkBuildParams &bp, bool sequential, int display_out[BK_MAX_PLATES])
{
    const size_t px_total = (size_t)ctx->W * ctx->rows();
    std::vector<uint32_t> off(px_total, BK_NULL_OFFSET);
    std::vector<uint8_t> tint(px_total, 255);
    int disp[BK_MAX_PLATES] = {0, 0, 0, 0, 0, 0};
    const HostTable T{&bp, off.data(), tint.data(), disp};
    const int ps = bp.ps, n1 = ps + 1;
    const size_t ncorner = (size_t)n1 * n1;
    std::vector<int> cx(ncorner), cy(ncorner), cerr(ncorner);
    std::vector<uint8_t> cok(ncorner), own((size_t)ps * ps, 1);
    const auto t0 = std::chrono::steady_clock::now();
    int errbits = 0;
    auto draw_row = [&](int plate, int py) {
        for (int px = 0; px < ps; ++px) {
            if (!own[(size_t)py * ps + px]) continue;                                               /* :2196 */
            const size_t c_tl = (size_t)py * n1 + px, c_bl = c_tl + n1;
            if (!(cok[c_tl] && cok[c_tl + 1] && cok[c_bl] && cok[c_bl + 1])) continue;
            const int tl[2] = {cx[c_tl], cy[c_tl]}, tr[2] = {cx[c_tl + 1], cy[c_tl + 1]}, bl[2] = {cx[c_bl], cy[c_bl]}, br[2] = {cx[c_bl + 1], cy[c_bl + 1]};
            h_draw_quad(T, tl, tr, bl, br, plate, px, py, h_offgrid(bp, px, py));
        }
    };
    try {
        const Values roots_in{P->lens_inverse, P->lens_forward, P->globe_plate};
        Values roots;
        std::unique_ptr<Interp> mine = P->interp.clone(roots_in, &roots);
        HostEval ev{mine.get(), roots[0], roots[1], roots[2]};
        for (int plate = 0; plate < bp.numplates && !errbits; ++plate) {
            const uint32_t c0 = (uint32_t)((size_t)plate * ncorner), t0id = (uint32_t)((size_t)plate * ps * ps);
            if (sequential) {
                auto corner_row = [&](int j) {
                    for (int i = 0; i < n1 && !errbits; ++i) {
                        const size_t k = (size_t)j * n1 + i;
                        h_corner_entry(ctx, ev, bp, c0 + (uint32_t)k, &cx[k], &cy[k], &cok[k], &errbits);
                    }
                };
                corner_row(ps);                                                                  /* :2148 the lower points of the last row */
                for (int py = ps - 1; py >= 0 && !errbits; --py) {
                    corner_row(py);                                                              /* :2172 upper points */
                    if (errbits) break;
                    for (int px = 0; px < ps; ++px) own[(size_t)py * ps + px] = h_texel_owns(ctx, ev, bp, t0id + (uint32_t)((size_t)py * ps + px)) ? 1 : 0;
                    draw_row(plate, py);
                }
            } else {
                std::fill(cerr.begin(), cerr.end(), 0);
                for_each_flagged(P, ncorner, [&](HostEval &E, size_t k) { h_corner_entry(ctx, E, bp, c0 + (uint32_t)k, &cx[k], &cy[k], &cok[k], &cerr[k]); });
                for (size_t k = 0; k < ncorner; ++k) errbits |= cerr[k];
                if (errbits) break;
                if (bp.has_globe_plate) for_each_flagged(P, (size_t)ps * ps, [&](HostEval &E, size_t k) { own[k] = h_texel_owns(ctx, E, bp, t0id + (uint32_t)k) ? 1 : 0; });
                else for (size_t k = 0; k < (size_t)ps * ps; ++k) own[k] = h_texel_owns(ctx, ev, bp, t0id + (uint32_t)k) ? 1 : 0;
                for (int py = ps - 1; py >= 0; --py) draw_row(plate, py);
            }
        }
    } catch (const LuaError &e) {
        return empty_host_build(ctx, display_out, e.what());
    }
    if (errbits) return empty_host_build(ctx, display_out, err_text(errbits));
    return finish_host_build(ctx, off, tint, disp, display_out, t0);
}

/* bk_build for a script the emitter declined (`why` = its refusal) */
static int build_on_host(bk_ctx *ctx, LensProgram *P, const std::string &why, int display_out[BK_MAX_PLATES])
{
    BkBuildParams bp;
    fill_params(ctx, &bp);
    bk::EmitRequest rq;
    rq.interp = &P->interp; rq.lens_inverse = P->lens_inverse; rq.lens_forward = P->lens_forward; rq.globe_plate = P->globe_plate;
    std::string which;
    bool carries = true;                                   // (a script the state analysis cannot follow either is taken to carry state)
    try { carries = bk::callbacks_carry_state(rq, &which); } catch (const LuaError &) { carries = true; which = "(callbacks not analysable)"; }
    const bool sequential = ctx->sequential_build >= 2 || (ctx->sequential_build == 1 && carries);
    ctx->last_build_path = sequential ? 2 : 1;
    ctx->last_build_why = why + (sequential ? (carries ? "; state carried in '" + which + "': one sequential scan on the host" : "; one sequential scan on the host (bk_set_sequential_build 2)")
                                            : "; callbacks carry no state: evaluated on the host's worker pool");
    if (P->info.map_type == BK_MAP_INVERSE) {
        if (!P->lens_inverse.is_function()) return ctx->fail(BK_E_STATE, "lens has no lens_inverse (map = \"lens_inverse\" without the function)");
        return sequential ? build_sequential(ctx, P, std::string(), bp, display_out) : build_inverse_pool(ctx, P, bp, display_out);
    }
    if (!P->lens_forward.is_function()) return ctx->fail(BK_E_STATE, "lens has no lens_forward");
    return build_forward_host(ctx, P, bp, sequential, display_out);
}

/* how the last bk_build evaluated the lens callbacks: 0 = the GPU kernels, 1 = the host's worker pool, 2 = one sequential scan on the
 * host; `why` (nullable) receives the reason for 1 / 2 - the construct the emitter declined, and the state finding */
extern "C" int bk_last_build_path(const bk_ctx *ctx, char *why, size_t cap)
{
    if (!ctx) return BK_E_INVALID;
    if (why && cap) snprintf(why, cap, "%s", ctx->last_build_why.c_str());
    return ctx->last_build_path;
}

#if BK_DEBUG_API
/* test hook: the HOST build paths on any context, a device-less one included (the CPU suite and the sanitizer run reach them this way).
 * mode 1 = the worker pool, 2 = one sequential scan, 0 = whichever bk_build would take for a script the emitter declines.  The table
 * comes back in the reference's layout (plate * ps * ps + py * ps + px, 0xFFFFFFFF = NULL); display_out as bk_build's; the return value
 * is bk_build's (BK_E_SCRIPT with the truncated / empty table in place).  Inverse maps never use the compiled host module here. */
extern "C" int bk_debug_host_build(bk_ctx *ctx, int mode, uint32_t *offsets, uint8_t *tints, int display_out[BK_MAX_PLATES], double *scale_out)
{
    if (!ctx || !offsets || !tints || mode < 0 || mode > 2) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "not a valid lens");
    if (!ctx->globe_valid) return ctx->fail(BK_E_STATE, "not a valid globe");
    if (ctx->W <= 0 || ctx->rows() <= 0) return ctx->fail(BK_E_STATE, "bk_debug_host_build: call bk_resize first");
    const size_t px = (size_t)ctx->W * ctx->rows();
    memset(offsets, 0xFF, px * 4);
    memset(tints, 255, px);
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = 0; if (display_out) display_out[i] = 0; }
    ctx->last_bad_key = 0;
    if (int r = bk_calc_zoom(ctx, scale_out)) return r;
    if (P->info.map_type == BK_MAP_NONE) return ctx->fail(BK_E_STATE, "no inverse or forward map being used");
    BkBuildParams bp;
    fill_params(ctx, &bp);
    ctx->host_sink_off = offsets; ctx->host_sink_tint = tints;
    int rc;
    if (mode == 0) rc = build_on_host(ctx, P, "bk_debug_host_build", display_out);
    else if (P->info.map_type == BK_MAP_INVERSE) {
        if (!P->lens_inverse.is_function()) rc = ctx->fail(BK_E_STATE, "lens has no lens_inverse");
        else rc = mode == 2 ? build_sequential(ctx, P, std::string(), bp, display_out) : build_inverse_pool(ctx, P, bp, display_out);
    } else {
        if (!P->lens_forward.is_function()) rc = ctx->fail(BK_E_STATE, "lens has no lens_forward");
        else rc = build_forward_host(ctx, P, bp, mode == 2, display_out);
    }
    ctx->host_sink_off = nullptr; ctx->host_sink_tint = nullptr;
    for (size_t i = 0; i < px; ++i) {
        if (offsets[i] == BK_NULL_OFFSET) continue;
        unsigned plate, x, y;
        bk_texel_coords((unsigned)ctx->gp, (unsigned)ctx->ph, offsets[i], &plate, &x, &y);
        offsets[i] = plate * (unsigned)(ctx->ps * ctx->ps) + y * (unsigned)ctx->ps + x;
    }
    return rc;
}
#endif

extern "C" int bk_set_sequential_build(bk_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return BK_E_INVALID;
    ctx->sequential_build = mode;
    return BK_OK;
}

/* 1 if a callback of the current lens / globe reads a script global that callbacks assign before assigning it itself (state can
 * travel from pixel to pixel), 0 if not, negative on error; `global_name` (nullable) receives the first such global */
extern "C" int bk_lens_carries_state(bk_ctx *ctx, char *global_name, size_t cap)
{
    if (!ctx) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "not a valid lens");
    bk::EmitRequest rq;
    rq.interp = &P->interp; rq.lens_inverse = P->lens_inverse; rq.lens_forward = P->lens_forward; rq.globe_plate = P->globe_plate;
    std::string which;
    bool carries = false;
    try { carries = bk::callbacks_carry_state(rq, &which); } catch (const LuaError &e) { return ctx->fail(BK_E_SCRIPT, "%s", e.what()); }
    if (global_name && cap) snprintf(global_name, cap, "%s", which.c_str());
    return carries ? 1 : 0;
}

extern "C" int bk_build(bk_ctx *ctx, int display_out[BK_MAX_PLATES], double *scale_out)
{
    if (!ctx) return BK_E_INVALID;
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "bk_build: this context has no device");
    if (!ctx->d_offsets) return ctx->fail(BK_E_STATE, "bk_build: call bk_resize first");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    bk::Range range("bk_build");
    LensProgram *P = ctx->prog;
    if (bk::build_module_ready(ctx) == BK_PENDING) return BK_PENDING;
    const size_t px = (size_t)ctx->W * ctx->rows();
    // F_RenderView clears the maps before (re)building, fisheye.c:731-732; whatever fails below,
    // the lensmap stays valid-and-empty so that bk_apply draws nothing, as the reference does.
    // (the device passes write every entry of the owned rows, so the clearing - 41 MB at 4K - is left to the ways out that have not
    //  filled the table: `empty_unless_built`; the host paths are handed a cleared table up front, as before)
    struct EmptyUnlessBuilt {
        bk_ctx *ctx; size_t px; bool armed = true;
        void now() { if (armed) { (void)hipMemsetAsync(ctx->d_offsets, 0xFF, px * 4, ctx->stream); (void)hipMemsetAsync(ctx->d_tints, 255, px, ctx->stream); armed = false; } }
        ~EmptyUnlessBuilt() { now(); }
    } empty_unless_built{ctx, px};
    BK_HIP(ctx, hipMemsetAsync(ctx->d_display, 0, 2 * (BK_MAX_PLATES + 3) * sizeof(int), ctx->stream));      // (two sets of counters: see the forward build)
    ctx->lensmap_valid = true;
    ctx->last_bad_key = 0;
    ctx->spans_valid = false;
    bk::coopmap_invalidate(ctx);
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = 0; if (display_out) display_out[i] = 0; }
    ctx->last_build_ms = 0;
    ctx->last_host_eval_ms = ctx->last_kernel_wall_ms = 0;
    ctx->last_kernel_retries = 0;
    ctx->last_fixup_compiled = false;
    ctx->last_flagged = ctx->last_changed = 0;
    ctx->last_build_path = 0;
    ctx->last_build_why.clear();
    ctx->fwd_tiles_used = false;

    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "not a valid lens");               /* create_lensmap :2372 */
    if (!ctx->globe_valid) return ctx->fail(BK_E_STATE, "not a valid globe");
    // Callbacks that were found to change nothing a script can see (LensProgram::emitted) leave the interpreter's activity count where
    // the translation unit was generated, however often this build runs them - calc_zoom, the flagged entries; a build runs nothing
    // else: the next bk_build of the same lens finds its translation unit still valid.
    struct KeepActivity {
        LensProgram *P; unsigned long long at; bool armed;
        ~KeepActivity() { if (P->emitted.valid && P->emitted.stateless && P->emitted.activity >= at) P->interp.activity = P->emitted.activity; }
    } keep_activity{P, P->interp.activity, P->emitted.valid && P->emitted.stateless && P->emitted.activity == P->interp.activity};
    if (int r = bk_calc_zoom(ctx, scale_out)) return r;                                        /* :2376 */
    if (P->info.map_type == BK_MAP_NONE) return ctx->fail(BK_E_STATE, "no inverse or forward map being used");   /* :2395 */

    std::string src, refused;
    if (keep_activity.armed) P->interp.activity = keep_activity.at;      // (calc_zoom ran the callbacks: they are stateless)
    if (int r = generate_source(ctx, P, &src, &refused)) return r;
    if (!refused.empty()) { empty_unless_built.now(); return build_on_host(ctx, P, refused, display_out); }     // callbacks the emitter declines: the interpreter evaluates them
    if (int r = compile_module(ctx, P, src)) return r;
    if (!ctx->d_flag_list) {
        BK_HIP(ctx, hipMalloc((void **)&ctx->d_flag_list, (size_t)65536 * 4 * sizeof(uint32_t)));
        ctx->flag_cap = 65536;
    }

    BkBuildParams bp;
    fill_params(ctx, &bp);
    // bk_set_sequential_build: a lens whose callbacks carry state from pixel to pixel (or every lens, mode 2) is built the way the
    // reference builds it - one evaluator, its scan order - on the host
    if ((P->info.map_type == BK_MAP_INVERSE || P->info.map_type == BK_MAP_FORWARD) && ctx->sequential_build) {
        bk::EmitRequest rq;
        rq.interp = &P->interp; rq.lens_inverse = P->lens_inverse; rq.lens_forward = P->lens_forward; rq.globe_plate = P->globe_plate;
        std::string which;
        if (ctx->sequential_build >= 2 || bk::callbacks_carry_state(rq, &which)) {
            ctx->last_build_path = 2;
            ctx->last_build_why = ctx->sequential_build >= 2 ? "bk_set_sequential_build 2" : "state carried in '" + which + "'";
            empty_unless_built.now();
            if (P->info.map_type == BK_MAP_INVERSE) return build_sequential(ctx, P, src, bp, display_out);
            // (r6) the forward scan is just as sequential in the reference (fisheye.c:2126-2217): its call order, one evaluator
            if (!P->lens_forward.is_function()) return ctx->fail(BK_E_STATE, "lens has no lens_forward");
            return build_forward_host(ctx, P, bp, true, display_out);
        }
    }
    void *args[] = {&bp};
    hipEvent_t e0, e1;
    for (hipEvent_t &e : ctx->build_time_ev) if (!e) BK_HIP(ctx, hipEventCreate(&e));
    e0 = ctx->build_time_ev[0]; e1 = ctx->build_time_ev[1];
    void *scratch[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = BK_OK;
    auto cleanup = [&]() {
        for (void *p : scratch) if (p) (void)hipFree(p);
    };
#define BK_HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = ctx->fail(BK_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
#define BK_RC_C(expr) do { rc = (expr); if (rc != BK_OK) { cleanup(); return rc; } } while (0)
    int flags[BK_MAX_PLATES + 3];
    unsigned int host_bad_key = 0;                     // (the same, among the entries the host re-derived)
    int host_display[BK_MAX_PLATES] = {0, 0, 0, 0, 0, 0};
    int host_err = 0;
    std::vector<uint32_t> flagged;
    auto reset_counters = [&]() -> hipError_t { return hipMemsetAsync(ctx->d_display, 0, (BK_MAX_PLATES + 3) * sizeof(int), ctx->stream); };
    if (!ctx->h_build_flags) BK_HIP_C(hipHostMalloc((void **)&ctx->h_build_flags, 2 * sizeof flags, hipHostMallocDefault));
    auto read_counters = [&]() -> hipError_t {       // (into pinned memory: a pageable destination goes through the runtime's staging buffer)
        hipError_t e = hipMemcpyAsync(ctx->h_build_flags, ctx->d_display, sizeof flags, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) memcpy(flags, ctx->h_build_flags, sizeof flags);
        return e;
    };

    try {
        if (P->info.map_type == BK_MAP_INVERSE) {
            if (!P->k_inverse) { cleanup(); return ctx->fail(BK_E_STATE, "lens has no lens_inverse (map = \"lens_inverse\" without the function)"); }
            for (int k = 0; k < 4; ++k) {                     // (a forward build's scratch is not kept under an inverse lens)
                (void)hipFree(ctx->fwd_scratch[k]);
                ctx->fwd_scratch[k] = nullptr; ctx->fwd_scratch_bytes[k] = 0;
            }
            BK_HIP_C(hipEventRecord(e0, ctx->stream));
            const auto tk0 = std::chrono::steady_clock::now();
            for (;;) {
                BK_HIP_C(hipModuleLaunchKernel(P->k_inverse, (unsigned)((ctx->W + 255) / 256), (unsigned)ctx->rows(), 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
                BK_HIP_C(read_counters());
                bool retry = false;
                BK_RC_C(read_flagged(ctx, (unsigned)flags[BK_MAX_PLATES + 1], &flagged, &retry));
                if (!retry) break;
                ++ctx->last_kernel_retries;
                fill_params(ctx, &bp);                       // (the list moved)
                BK_HIP_C(reset_counters());
            }
            ctx->last_kernel_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk0).count();
            // re-derive the flagged pixels on the host and patch the ones that differ
            std::vector<uint32_t> idx, voff;
            std::vector<uint8_t> vtint;
            const size_t nfl = flagged.size() / 4;
            {
                std::vector<uint32_t> roff(nfl);
                std::vector<uint8_t> rtint(nfl);
                std::vector<int> rshown(nfl), rerr(nfl, 0);
                const auto th0 = std::chrono::steady_clock::now();
                const HostModuleP hm = nfl ? fixup_module(P, src) : nullptr;
                ctx->last_fixup_compiled = hm != nullptr;
                if (hm) hostmod_runs(nfl, [&](size_t i0, size_t cnt) {
                    hm->inverse(&bp, &flagged[4 * i0], 4, (unsigned long)cnt, &roff[i0], &rtint[i0], &rshown[i0], &rerr[i0]);
                });
                else for_each_flagged(P, nfl, [&](HostEval &E, size_t i) {
                    h_inverse_entry(E, bp, flagged[4 * i], &roff[i], &rtint[i], &rshown[i], &rerr[i]);
                });
                ctx->last_host_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
                for (size_t i = 0; i < nfl; ++i) {
                    host_err |= rerr[i];
                    if (rerr[i] & BK_ERR_RESULT) {
                        const uint32_t o = flagged[4 * i], lyl = o / (uint32_t)ctx->W, lx = o - lyl * (uint32_t)ctx->W;
                        host_bad_key = std::max(host_bad_key, (uint32_t)(((uint32_t)ctx->row0 + lyl) * (uint32_t)ctx->W + ((uint32_t)ctx->W - 1u - lx)) + 1u);
                    }
                    if (rshown[i] >= 0) host_display[rshown[i]] = 1;
                    if (roff[i] != flagged[4 * i + 1] || rtint[i] != (uint8_t)flagged[4 * i + 2]) {
                        idx.push_back(flagged[4 * i]); voff.push_back(roff[i]); vtint.push_back(rtint[i]);
                    }
                }
            }
            ctx->last_flagged = (int)nfl;
            ctx->last_changed = (int)idx.size();
            BK_RC_C(bk::launch_scatter32(ctx, ctx->d_offsets, idx.data(), voff.data(), idx.size()));
            BK_RC_C(bk::launch_scatter8(ctx, ctx->d_tints, idx.data(), vtint.data(), idx.size()));
            BK_HIP_C(hipEventRecord(e1, ctx->stream));
        } else {
            if (!P->k_corners || !P->k_quads || !P->k_resolve) { cleanup(); return ctx->fail(BK_E_STATE, "lens has no lens_forward"); }
            const size_t n1 = (size_t)ctx->ps + 1;
            const size_t ncorner = (size_t)ctx->numplates * n1 * n1;
            const size_t want[4] = {ncorner * 2 * sizeof(int), ncorner, px * 4, px * 4};
            for (int k = 0; k < 4; ++k)
                if (ctx->fwd_scratch_bytes[k] < want[k]) {
                    (void)hipFree(ctx->fwd_scratch[k]);
                    ctx->fwd_scratch[k] = nullptr; ctx->fwd_scratch_bytes[k] = 0;
                    BK_HIP_C(hipMalloc(&ctx->fwd_scratch[k], want[k]));
                    ctx->fwd_scratch_bytes[k] = want[k];
                }
            bp.corner_xy = (int *)ctx->fwd_scratch[0];
            bp.corner_ok = (unsigned char *)ctx->fwd_scratch[1];
            bp.fwd_key_px = (unsigned int *)ctx->fwd_scratch[2];
            bp.fwd_key_tint = (unsigned int *)ctx->fwd_scratch[3];
            // the quotient / uv tables of bk_build_params.h: plain IEEE divisions, done here once per platesize instead of per texel
            constexpr size_t NQ = 21 * 21;
            const size_t ntile = ((size_t)ctx->ps + 15) / 16;              // (BK_FWD_TILE = 16; the tile flags of bk_forward_tiles live behind the tables)
            if (ctx->fwd_tables_ps != ctx->ps) {
                (void)hipFree(ctx->fwd_tables);
                ctx->fwd_tables = nullptr; ctx->fwd_tables_ps = -1;
                std::vector<double> q(NQ, 0.0);
                for (int a = 0; a <= 20; ++a)
                    for (int d = 1; d <= 20; ++d) q[(size_t)a * 21 + d] = (double)a / (double)d;
                std::vector<float> uv(2 * n1);
                for (size_t i = 0; i < n1; ++i) {
                    uv[i] = (float)(((double)i - 0.5) / ctx->ps - 0.5);
                    uv[n1 + i] = (float)((double)i / ctx->ps - 0.5);
                }
                BK_HIP_C(hipMalloc(&ctx->fwd_tables, NQ * sizeof(double) + uv.size() * sizeof(float) + (size_t)BK_MAX_PLATES * ntile * ntile));
                BK_HIP_C(hipMemcpy(ctx->fwd_tables, q.data(), NQ * sizeof(double), hipMemcpyHostToDevice));
                BK_HIP_C(hipMemcpy((char *)ctx->fwd_tables + NQ * sizeof(double), uv.data(), uv.size() * sizeof(float), hipMemcpyHostToDevice));
                ctx->fwd_tables_ps = ctx->ps;
            }
            bp.fwd_quot = (const double *)ctx->fwd_tables;
            bp.fwd_uv = (const float *)((const char *)ctx->fwd_tables + NQ * sizeof(double));
            unsigned char *const tile_own = (unsigned char *)ctx->fwd_tables + NQ * sizeof(double) + 2 * n1 * sizeof(float);
            const auto clear_keys = [&](hipStream_t st) -> hipError_t {
                hipError_t e = hipMemsetAsync(ctx->fwd_scratch[2], 0, px * 4, st);
                return e == hipSuccess ? hipMemsetAsync(ctx->fwd_scratch[3], 0, px * 4, st) : e;
            };
            const auto launch_quads = [&](bool keys_cleared = false, void **with = nullptr) -> hipError_t {
                hipError_t e = keys_cleared ? hipSuccess : clear_keys(ctx->stream);
                if (e == hipSuccess) e = hipModuleLaunchKernel(P->k_quads, (unsigned)((ctx->ps + 15) / 16), (unsigned)((ctx->ps + 15) / 16), (unsigned)ctx->numplates, 256, 1, 1, 0,
                                                               ctx->stream, with ? with : args, nullptr);       // (BK_FWD_TILE = 16: bk_build_kernels.h)
                return e;
            };
            const auto launch_resolve = [&](void **with = nullptr) -> hipError_t {
                return hipModuleLaunchKernel(P->k_resolve, (unsigned)((px + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream, with ? with : args, nullptr);
            };
            // The three passes in one go, each of the first two counting into its own set of counters, both sets left in pinned memory
            // by the last pass: nearly every build flags nothing - no corner and no texel for the host
            // to look at again - and then the table is final when the stream drains, two stops (a read-back and a decision each, 25-30 us
            // apiece at 4K) earlier.  A build that did flag something is done over the careful way below, and so is the next build of
            // the same lens.
            bool speculated = false;
            if (!P->fwd_needs_host && !bk::g_debug.forward_careful) {
                constexpr size_t NF = BK_MAX_PLATES + 3;
                int *const after_corners = ctx->h_build_flags, *const after_quads = ctx->h_build_flags + NF;
                if (!ctx->build_aux) {
                    BK_HIP_C(hipStreamCreateWithFlags(&ctx->build_aux, hipStreamNonBlocking));
                    for (hipEvent_t &e : ctx->build_ev) BK_HIP_C(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                }
                BK_HIP_C(hipEventRecord(e0, ctx->stream));
                BK_HIP_C(hipEventRecord(ctx->build_ev[0], ctx->stream));
                BK_HIP_C(hipModuleLaunchKernel(P->k_corners, (unsigned)((n1 + 255) / 256), (unsigned)n1, (unsigned)ctx->numplates, 256, 1, 1, 0, ctx->stream, args, nullptr));
                // the key planes (66 MB at 4K) are cleared on the side stream while the corner pass - arithmetic only - has the chip
                // (after whatever the stream held before this build, which may still read them: build_ev[0])
                BK_HIP_C(hipStreamWaitEvent(ctx->build_aux, ctx->build_ev[0], 0));
                BK_HIP_C(clear_keys(ctx->build_aux));
                // ... and the tiles that lie wholly inside their plate's own region are found there too (the plates may have changed
                // since the last build: 110 K threads, a few microseconds)
                BkBuildParams bq = bp;
                if (P->k_tiles) {
                    unsigned char *flags_out = tile_own;
                    void *args_t[] = {&bp, &flags_out};
                    BK_HIP_C(hipModuleLaunchKernel(P->k_tiles, (unsigned)((ntile * ntile + 255) / 256), (unsigned)ctx->numplates, 1, 256, 1, 1, 0, ctx->build_aux, args_t, nullptr));
                    bq.tile_own = tile_own;
                    ctx->fwd_tiles_used = true;
                }
                BK_HIP_C(hipEventRecord(ctx->build_ev[1], ctx->build_aux));
                bq.display = ctx->d_display + NF;            // the quad pass counts into the second set (cleared with the first, above)
                bq.err = bq.display + BK_MAX_PLATES;
                bq.flag_count = (unsigned int *)(bq.display + BK_MAX_PLATES + 1);
                bq.first_bad = (unsigned int *)(bq.display + BK_MAX_PLATES + 2);
                void *args_q[] = {&bq};
                BK_HIP_C(hipStreamWaitEvent(ctx->stream, ctx->build_ev[1], 0));
                BK_HIP_C(launch_quads(true, args_q));
                static_assert(2 * NF == 18, "bk_forward_resolve copies 18 counters");
                BkBuildParams br = bp;
                br.counters_out = ctx->h_build_flags;
                void *args_r[] = {&br};
                BK_HIP_C(launch_resolve(args_r));
                BK_HIP_C(hipEventRecord(e1, ctx->stream));
                BK_HIP_C(hipStreamSynchronize(ctx->stream));
                if (after_corners[BK_MAX_PLATES + 1] == 0 && after_quads[BK_MAX_PLATES + 1] == 0) {
                    memcpy(flags, after_quads, sizeof flags);
                    flags[BK_MAX_PLATES] |= after_corners[BK_MAX_PLATES];
                    ctx->last_flagged = 0;
                    ctx->last_changed = 0;
                    speculated = true;
                } else {
                    P->fwd_needs_host = true;
                    ctx->fwd_tiles_used = false;
                    BK_HIP_C(reset_counters());
                }
            }
            if (!speculated) {
                BK_HIP_C(hipEventRecord(e0, ctx->stream));
                // texel corners -> screen; the flagged ones re-derived on the host
                for (;;) {
                    BK_HIP_C(hipModuleLaunchKernel(P->k_corners, (unsigned)((n1 + 255) / 256), (unsigned)n1, (unsigned)ctx->numplates, 256, 1, 1, 0, ctx->stream, args, nullptr));
                    BK_HIP_C(read_counters());
                    bool retry = false;
                    BK_RC_C(read_flagged(ctx, (unsigned)flags[BK_MAX_PLATES + 1], &flagged, &retry));
                    if (!retry) break;
                    bp.flag_list = ctx->d_flag_list; bp.flag_cap = (unsigned)ctx->flag_cap;
                    BK_HIP_C(reset_counters());
                }
                int corner_err = flags[BK_MAX_PLATES];
                {
                    std::vector<uint32_t> ixy, vxy, iok;
                    std::vector<uint8_t> vok;
                    const size_t nfl = flagged.size() / 4;
                    std::vector<int> rsx(nfl), rsy(nfl), rerr(nfl, 0);
                    std::vector<uint8_t> rok(nfl);
                    const auto th0 = std::chrono::steady_clock::now();
                    const HostModuleP hm = nfl ? fixup_module(P, src) : nullptr;
                    ctx->last_fixup_compiled = hm != nullptr;
                    if (hm) hostmod_runs(nfl, [&](size_t i0, size_t cnt) {
                        hm->corners(&bp, &flagged[4 * i0], 4, (unsigned long)cnt, &rsx[i0], &rsy[i0], &rok[i0], &rerr[i0]);
                    });
                    else for_each_flagged(P, nfl, [&](HostEval &E, size_t i) {
                        h_corner_entry(ctx, E, bp, flagged[4 * i], &rsx[i], &rsy[i], &rok[i], &rerr[i]);
                    });
                    ctx->last_host_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
                    for (size_t i = 0; i < nfl; ++i) {
                        const size_t k = 4 * i;
                        const int sx = rsx[i], sy = rsy[i];
                        const uint8_t ok = rok[i];
                        host_err |= rerr[i];
                        if ((uint32_t)sx != flagged[k + 1] || (uint32_t)sy != flagged[k + 2] || ok != (uint8_t)flagged[k + 3]) {
                            ixy.push_back(2 * flagged[k]); vxy.push_back((uint32_t)sx);
                            ixy.push_back(2 * flagged[k] + 1); vxy.push_back((uint32_t)sy);
                            iok.push_back(flagged[k]); vok.push_back(ok);
                        }
                    }
                    ctx->last_flagged = (int)(flagged.size() / 4);
                    ctx->last_changed = (int)iok.size();
                    BK_RC_C(bk::launch_scatter32(ctx, (uint32_t *)bp.corner_xy, ixy.data(), vxy.data(), ixy.size()));
                    BK_RC_C(bk::launch_scatter8(ctx, bp.corner_ok, iok.data(), vok.data(), iok.size()));
                }
                // quads; with a globe_plate script a texel's "own plate" test can be flagged too: the host answers those and
                // the scatter runs once more with its answers
                std::vector<uint32_t> ovr;
                for (int pass = 0; pass < 2; ++pass) {
                    bool again = false;
                    for (;;) {
                        BK_HIP_C(reset_counters());
                        BK_HIP_C(launch_quads());
                        BK_HIP_C(read_counters());
                        if (pass == 1) break;
                        bool retry = false;
                        BK_RC_C(read_flagged(ctx, (unsigned)flags[BK_MAX_PLATES + 1], &flagged, &retry));
                        if (!retry) break;
                        bp.flag_list = ctx->d_flag_list; bp.flag_cap = (unsigned)ctx->flag_cap;
                    }
                    if (pass == 0 && !flagged.empty()) {
                        std::vector<std::pair<uint32_t, uint32_t>> ans;
                        const size_t nfl = flagged.size() / 4;
                        std::vector<uint8_t> rown(nfl);
                        const HostModuleP hm = fixup_module(P, src);
                        if (hm) hostmod_runs(nfl, [&](size_t i0, size_t cnt) { hm->texel_owns(&bp, &flagged[4 * i0], 4, (unsigned long)cnt, &rown[i0]); });
                        else for_each_flagged(P, nfl, [&](HostEval &E, size_t i) { rown[i] = h_texel_owns(ctx, E, bp, flagged[4 * i]) ? 1 : 0; });
                        for (size_t k = 0; k + 3 < flagged.size(); k += 4) {
                            const bool own = rown[k / 4] != 0;
                            if ((own ? 1u : 0u) != flagged[k + 1]) { again = true; ++ctx->last_changed; }
                            ans.push_back({flagged[k], own ? 1u : 0u});
                        }
                        ctx->last_flagged += (int)ans.size();
                        if (again) {
                            std::sort(ans.begin(), ans.end());
                            for (auto &a : ans) { ovr.push_back((a.first << 1) | a.second); }
                            BK_HIP_C(hipMalloc(&scratch[4], ovr.size() * 4));
                            BK_HIP_C(hipMemcpyAsync(scratch[4], ovr.data(), ovr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
                            bp.ovr_list = (const unsigned int *)scratch[4];
                            bp.ovr_count = (unsigned)ovr.size();
                        }
                    }
                    if (!again) break;
                }
                flags[BK_MAX_PLATES] |= corner_err;
                BK_HIP_C(launch_resolve());
                BK_HIP_C(hipEventRecord(e1, ctx->stream));
                P->fwd_needs_host = ctx->last_flagged != 0;
            }
        }
    } catch (const LuaError &e) {
        cleanup();
        return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", e.what());
    }
    BK_HIP_C(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ctx->last_build_ms = ms;
    cleanup();
#undef BK_HIP_C
#undef BK_RC_C
    for (int i = 0; i < BK_MAX_PLATES; ++i) {
        ctx->display[i] = i < ctx->numplates ? (flags[i] | host_display[i]) : 0;
        if (display_out) display_out[i] = ctx->display[i];
    }
    const int errbits = flags[BK_MAX_PLATES] | host_err;
    if (errbits == BK_ERR_RESULT && P->info.map_type == BK_MAP_INVERSE) {
        // A malformed callback result (status -1) ends the reference's scan at that pixel and KEEPS what it had set by then
        // (fisheye.c:2113-2117; rows from the bottom up, pixels left to right; run to completion, no time slicing).  The GPU
        // build has evaluated every pixel: take away what the reference had not reached, recount the display flags, and
        // report the error with that table in place.  (A stripe context knows only its own rows: bk_multi_build / the host
        // of a bk_comm group hands every stripe the group's first failing pixel - bk_truncate_build.)
        ctx->last_bad_key = std::max((unsigned int)flags[BK_MAX_PLATES + 2], host_bad_key);
        int disp[BK_MAX_PLATES];
        if (int r = bk::launch_truncate_scan(ctx, ctx->last_bad_key, disp)) return r;
        for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = i < ctx->numplates ? disp[i] : 0; if (display_out) display_out[i] = ctx->display[i]; }
        empty_unless_built.armed = false;
        return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", err_text(errbits));
    }
    if (errbits) {
        // Any other per-pixel error is a Lua runtime error, which the reference does not survive (lua_call is unprotected:
        // fisheye.c:1551); here the build leaves an EMPTY map (nothing is drawn) and reports it.
        empty_unless_built.now();
        BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = 0; if (display_out) display_out[i] = 0; }
        return ctx->fail(BK_E_SCRIPT, "lensmap build: %s", err_text(errbits));
    }
    empty_unless_built.armed = false;
    return BK_OK;
}

#if BK_DEBUG_API
extern "C" int bk_debug_forward_tiles(bk_ctx *ctx, int *taken, int *total)
{
    if (!ctx || !taken || !total) return BK_E_INVALID;
    const size_t nt = ((size_t)ctx->ps + 15) / 16, n = (size_t)ctx->numplates * nt * nt;
    *total = (int)n;
    *taken = -1;
    if (!ctx->fwd_tiles_used || !ctx->fwd_tables || ctx->fwd_tables_ps != ctx->ps) return BK_OK;
    std::vector<unsigned char> flags(n);
    const size_t at = 21 * 21 * sizeof(double) + 2 * ((size_t)ctx->ps + 1) * sizeof(float);       // (behind the quotient and uv tables: bk_build)
    BK_HIP(ctx, hipSetDevice(ctx->device));
    BK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BK_HIP(ctx, hipMemcpy(flags.data(), (const char *)ctx->fwd_tables + at, n, hipMemcpyDeviceToHost));
    int k = 0;
    for (unsigned char f : flags) k += f ? 1 : 0;
    *taken = k;
    return BK_OK;
}
#endif

#if BK_DEBUG_API
/* test hook: the kernel-argument block bk_build would launch with (calc_zoom done, device pointers as they are -
 * null on a BK_DEVICE_NONE context).  tests/hostemu compiles the generated translation unit for the host and runs it
 * on this block to inspect the device code's results and flags without a GPU. */
extern "C" int bk_debug_build_params(bk_ctx *ctx, void *out, size_t cap, size_t *needed)
{
    if (!ctx) return BK_E_INVALID;
    if (needed) *needed = sizeof(BkBuildParams);
    if (!out) return BK_OK;
    if (cap < sizeof(BkBuildParams)) return ctx->fail(BK_E_INVALID, "bk_debug_build_params: buffer too small");
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "not a valid lens");
    if (!ctx->globe_valid) return ctx->fail(BK_E_STATE, "not a valid globe");
    if (int r = bk_calc_zoom(ctx, nullptr)) return r;
    BkBuildParams bp;
    fill_params(ctx, &bp);
    memcpy(out, &bp, sizeof bp);
    return BK_OK;
}
#endif

#if BK_DEBUG_API
/* test hook: the host re-evaluation bk_build applies to flagged pixels, run over ANY pixel indices of the owned rows
 * (works on a BK_DEVICE_NONE context).  offsets come back in the reference layout plate*ps*ps + py*ps + px. */
extern "C" int bk_debug_host_entries(bk_ctx *ctx, const uint32_t *ids, size_t n, uint32_t *offsets, uint8_t *tints)
{
    if (!ctx || !ids || !offsets || !tints) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid || !P->lens_inverse.is_function()) return ctx->fail(BK_E_STATE, "no lens_inverse");
    if (!ctx->globe_valid) return ctx->fail(BK_E_STATE, "not a valid globe");
    if (int r = bk_calc_zoom(ctx, nullptr)) return r;
    BkBuildParams bp;
    fill_params(ctx, &bp);
    const size_t px = (size_t)ctx->W * ctx->rows();
    for (size_t i = 0; i < n; ++i) if (ids[i] >= px) return ctx->fail(BK_E_INVALID, "bk_debug_host_entries: index out of range");
    std::vector<int> shown(n), err(n, 0);
    try {
        std::string src;
        HostModuleP hm;
        if (bk::g_debug.host_module != 2 && generate_source(ctx, P, &src) == BK_OK) hm = fixup_module(P, src);
        if (bk::g_debug.host_module == 1 && !hm) return ctx->fail(BK_E_STATE, "bk_debug_host_entries: no host module (no C++ compiler, or host math is not the platform libm)");
        if (hm) hostmod_runs(n, [&](size_t i0, size_t cnt) { hm->inverse(&bp, &ids[i0], 1, (unsigned long)cnt, &offsets[i0], &tints[i0], &shown[i0], &err[i0]); });
        else for_each_flagged(P, n, [&](HostEval &E, size_t i) { h_inverse_entry(E, bp, ids[i], &offsets[i], &tints[i], &shown[i], &err[i]); });
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "%s", e.what());
    }
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i] == BK_NULL_OFFSET) continue;
        unsigned plate, x, y;
        bk_texel_coords((unsigned)ctx->gp, (unsigned)ctx->ph, offsets[i], &plate, &x, &y);
        offsets[i] = plate * (unsigned)(ctx->ps * ctx->ps) + y * (unsigned)ctx->ps + x;
    }
    return BK_OK;
}
#endif

#if BK_DEBUG_API
/* the same for the forward build's texel corners (ids: corner number plate * (ps+1)^2 + j * (ps+1) + i): screen x, y and
 * whether lens_forward gave a position */
extern "C" int bk_debug_host_corners(bk_ctx *ctx, const uint32_t *ids, size_t n, int32_t *sx, int32_t *sy, uint8_t *ok)
{
    if (!ctx || !ids || !sx || !sy || !ok) return BK_E_INVALID;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid || !P->lens_forward.is_function()) return ctx->fail(BK_E_STATE, "no lens_forward");
    if (!ctx->globe_valid) return ctx->fail(BK_E_STATE, "not a valid globe");
    if (int r = bk_calc_zoom(ctx, nullptr)) return r;
    BkBuildParams bp;
    fill_params(ctx, &bp);
    const size_t n1 = (size_t)ctx->ps + 1, total = (size_t)ctx->numplates * n1 * n1;
    for (size_t i = 0; i < n; ++i) if (ids[i] >= total) return ctx->fail(BK_E_INVALID, "bk_debug_host_corners: index out of range");
    std::vector<int> err(n, 0), x(n), y(n);
    try {
        std::string src;
        HostModuleP hm;
        if (bk::g_debug.host_module != 2 && generate_source(ctx, P, &src) == BK_OK) hm = fixup_module(P, src);
        if (bk::g_debug.host_module == 1 && !hm) return ctx->fail(BK_E_STATE, "bk_debug_host_corners: no host module (no C++ compiler, or host math is not the platform libm)");
        if (hm) hostmod_runs(n, [&](size_t i0, size_t cnt) { hm->corners(&bp, &ids[i0], 1, (unsigned long)cnt, &x[i0], &y[i0], &ok[i0], &err[i0]); });
        else for_each_flagged(P, n, [&](HostEval &E, size_t i) { h_corner_entry(ctx, E, bp, ids[i], &x[i], &y[i], &ok[i], &err[i]); });
    } catch (const LuaError &e) {
        return ctx->fail(BK_E_SCRIPT, "%s", e.what());
    }
    for (size_t i = 0; i < n; ++i) { sx[i] = x[i]; sy[i] = y[i]; }
    return BK_OK;
}
#endif

#if BK_DEBUG_API
extern "C" int bk_debug_build_breakdown(const bk_ctx *ctx, double out[6])
{
    if (!ctx || !out) return BK_E_INVALID;
    out[0] = ctx->last_build_ms;
    out[1] = ctx->last_host_eval_ms;
    out[2] = (double)ctx->last_flagged;
    out[3] = (double)FixupPool::get().size();
    out[4] = ctx->last_kernel_wall_ms;
    out[5] = (double)ctx->last_kernel_retries + (ctx->last_fixup_compiled ? 1000.0 : 0.0);
    return BK_OK;
}
#endif

/* Is the compiled host module of the current lens + globe there (see bk_set_host_compile)?  wait != 0: compile it now if need
 * be and wait for it.  1 = ready, 0 = not (still compiling, no compiler, switched off, host math not the platform libm). */
extern "C" int bk_host_module_ready(bk_ctx *ctx, int wait)
{
    if (!ctx) return 0;
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid || !ctx->globe_valid || P->info.map_type == BK_MAP_NONE) return 0;
    if (P->interp.math != &math_platform()) return 0;
    std::string src;
    if (generate_source(ctx, P, &src) != BK_OK) return 0;
    return host_module_for(src, wait != 0) ? 1 : 0;
}

/* Multi-GPU companion of bk_build's handling of a malformed callback result: `bad_key` = the group's first failing pixel
 * (max over the stripes of bk_last_build_bad_key); this stripe gives up what the reference's scan had not reached by then. */
extern "C" unsigned int bk_last_build_bad_key(const bk_ctx *ctx) { return ctx ? ctx->last_bad_key : 0u; }
extern "C" int bk_truncate_build(bk_ctx *ctx, unsigned int bad_key, int display_out[BK_MAX_PLATES])
{
    if (!ctx) return BK_E_INVALID;
    if (!ctx->lensmap_valid || !ctx->d_offsets) return ctx->fail(BK_E_STATE, "bk_truncate_build: no lensmap");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    int disp[BK_MAX_PLATES];
    if (bad_key == 0) return BK_OK;
    if (int r = bk::launch_truncate_scan(ctx, bad_key, disp)) return r;
    ctx->last_bad_key = bad_key;
    ctx->spans_valid = false;
    bk::coopmap_invalidate(ctx);
    for (int i = 0; i < BK_MAX_PLATES; ++i) { ctx->display[i] = i < ctx->numplates ? disp[i] : 0; if (display_out) display_out[i] = ctx->display[i]; }
    return BK_OK;
}

extern "C" int bk_last_build_fixups(const bk_ctx *ctx, int *flagged, int *changed)
{
    if (!ctx) return BK_E_INVALID;
    if (flagged) *flagged = ctx->last_flagged;
    if (changed) *changed = ctx->last_changed;
    return BK_OK;
}

// debug / test hook: run a callback on the DEVICE over n argument tuples (nargs doubles each);
// out receives 8 doubles per tuple, nout the result count (-1 single nil, <= -100 runtime error)
/* f_saveglobe's plate image (fisheye.c:1438-1456), computed on the device; the caller encodes the PCX */
extern "C" int bk_save_plate(bk_ctx *ctx, int frame, int plate, int with_margins, uint8_t *dst_host, int dst_pitch)
{
    if (!ctx || !dst_host) return BK_E_INVALID;
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "bk_save_plate: this context has no device");
    if (!ctx->d_globe) return ctx->fail(BK_E_STATE, "bk_save_plate: call bk_resize first");
    if (!ctx->globe_valid || !ctx->prog) return ctx->fail(BK_E_STATE, "not a valid globe");
    if (plate < 0 || plate >= ctx->numplates || frame < 0 || frame >= ctx->nframes || dst_pitch < ctx->ps)
        return ctx->fail(BK_E_INVALID, "bk_save_plate: bad frame/plate/pitch");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    LensProgram *P = ctx->prog;
    std::string src;
    if (int r = generate_source(ctx, P, &src)) return r;
    if (int r = compile_module(ctx, P, src)) return r;
    hipFunction_t fn = nullptr;
    BK_HIP(ctx, hipModuleGetFunction(&fn, P->module, "bk_save_plate"));
    if (!ctx->d_flag_list) {
        BK_HIP(ctx, hipMalloc((void **)&ctx->d_flag_list, (size_t)65536 * 4 * sizeof(uint32_t)));
        ctx->flag_cap = 65536;
    }
    const uint8_t *globe = ctx->d_globe + (size_t)frame * ctx->globe_stride();
    uint8_t *out = nullptr;
    BK_HIP(ctx, hipMalloc((void **)&out, (size_t)ctx->ps * ctx->ps));
    std::vector<uint32_t> flagged;
    hipError_t e = hipSuccess;
    for (;;) {
        BkBuildParams bp;
        fill_params(ctx, &bp);
        void *args[] = {&bp, &plate, &with_margins, &globe, &out};
        int counters[BK_MAX_PLATES + 2];
        e = hipMemsetAsync(ctx->d_display, 0, sizeof counters, ctx->stream);
        if (e == hipSuccess) e = hipModuleLaunchKernel(fn, (unsigned)((ctx->ps + 255) / 256), (unsigned)ctx->ps, 1, 256, 1, 1, 0, ctx->stream, args, nullptr);
        if (e == hipSuccess) e = hipMemcpyAsync(counters, ctx->d_display, sizeof counters, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) break;
        bool retry = false;
        if (int r = read_flagged(ctx, (unsigned)counters[BK_MAX_PLATES + 1], &flagged, &retry)) { (void)hipFree(out); return r; }
        if (!retry) break;
    }
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(dst_host, (size_t)dst_pitch, out, (size_t)ctx->ps, (size_t)ctx->ps, (size_t)ctx->ps, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(out);
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "bk_save_plate failed: %s", hipGetErrorString(e));
    // texels whose plate ownership a globe_plate script decides on libm's last bits: the host interpreter has the say
    if (!flagged.empty()) {
        BkBuildParams bp;
        fill_params(ctx, &bp);
        const size_t nfl = flagged.size() / 4;
        std::vector<uint8_t> own(nfl);
        try {
            for_each_flagged(P, nfl, [&](HostEval &E, size_t k) {
                const uint32_t id = flagged[4 * k], i = id / (uint32_t)ctx->ps, j = id - i * (uint32_t)ctx->ps;
                float ray[3];
                bk::h_plate_uv_to_ray(ctx->plates[plate], (double)j / ctx->ps, (double)i / ctx->ps, ray);
                own[k] = plate == h_ray_to_plate_index(E, bp, ray);
            });
        } catch (const LuaError &err) {
            return ctx->fail(BK_E_SCRIPT, "%s", err.what());
        }
        for (size_t k = 0; k < nfl; ++k) {
            const uint32_t id = flagged[4 * k], i = id / (uint32_t)ctx->ps, j = id - i * (uint32_t)ctx->ps;
            dst_host[(size_t)i * dst_pitch + j] = own[k] ? (uint8_t)flagged[4 * k + 1] : 0xFE;
        }
    }
    return BK_OK;
}

#if BK_DEBUG_API
extern "C" int bk_debug_eval_device(bk_ctx *ctx, int which, const double *args, int nargs, int n, double *out, int *nout)
{
    if (!ctx || !args || !out || !nout || nargs < 1 || nargs > 4 || n < 1) return BK_E_INVALID;
    if (ctx->device < 0) return ctx->fail(BK_E_STATE, "this context has no device");
    LensProgram *P = ctx->prog;
    if (!P || !P->lens_valid) return ctx->fail(BK_E_STATE, "no valid lens");
    BK_HIP(ctx, hipSetDevice(ctx->device));
    bk::resident_quiesce(ctx);
    std::string src;
    if (int r = generate_source(ctx, P, &src)) return r;
    if (int r = compile_module(ctx, P, src)) return r;
    hipFunction_t fn = nullptr;
    BK_HIP(ctx, hipModuleGetFunction(&fn, P->module, "bk_eval_callback"));
    BkBuildParams bp;
    fill_params(ctx, &bp);
    double *d_args = nullptr, *d_out = nullptr;
    int *d_nout = nullptr;
    BK_HIP(ctx, hipMalloc((void **)&d_args, sizeof(double) * nargs * n));
    BK_HIP(ctx, hipMalloc((void **)&d_out, sizeof(double) * 8 * n));
    BK_HIP(ctx, hipMalloc((void **)&d_nout, sizeof(int) * n));
    BK_HIP(ctx, hipMemcpyAsync(d_args, args, sizeof(double) * nargs * n, hipMemcpyHostToDevice, ctx->stream));
    void *kargs[] = {&bp, &which, &d_args, &nargs, &n, &d_out, &d_nout};
    hipError_t e = hipModuleLaunchKernel(fn, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream, kargs, nullptr);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(double) * 8 * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(nout, d_nout, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_args); (void)hipFree(d_out); (void)hipFree(d_nout);
    if (e != hipSuccess) return ctx->fail(BK_E_HIP, "bk_debug_eval_device: %s", hipGetErrorString(e));
    return BK_OK;
}
#endif
