/*
 * oracle.c -- CPU ORACLE (test infrastructure, never linked into the product).
 *
 * Plain-C restatement of the reference lensmap build + apply, written from the
 * behaviour of engine/NQ/fisheye.c; see oracle.h for the parity status.
 * Compile with:  gcc -std=gnu99 -O2 -fwrapv -ffp-contract=off  (no -ffast-math, no -march)
 * so that float expressions are evaluated in float with no FMA contraction, as
 * on the reference's x86-64 build.
 */
#include "oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846   /* include/mathlib.h:56-57 */
#endif

#ifdef OK_PORTABLE_LIBM
/* liboracle_bkm.so: same algorithm, libm calls routed to the portable functions the GPU kernels
 * are built from.  Lets the tests separate "algorithm differs" from "platform libm last bit". */
#include "../blinky_amd/csrc/bkm.h"
#define sin bkm_sin
#define cos bkm_cos
#define tan bkm_tan
#define atan2 bkm_atan2
#define sqrt bkm_sqrt
#define fmod bkm_fmod
#endif

/* ---- mathlib.c restatements ------------------------------------------------- */

/* include/mathlib.h:70  DotProduct macro on vec_t=float operands */
static float dot3f(const float a[3], const float b[3])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* common/mathlib.c:350-355  VectorMA: scale is narrowed to float by the prototype */
static void vector_ma(const float a[3], float scale, const float b[3], float c[3])
{
    c[0] = a[0] + scale * b[0];
    c[1] = a[1] + scale * b[1];
    c[2] = a[2] + scale * b[2];
}

/* common/mathlib.c:389-394 */
static void cross3f(const float v1[3], const float v2[3], float cross[3])
{
    cross[0] = v1[1] * v2[2] - v1[2] * v2[1];
    cross[1] = v1[2] * v2[0] - v1[0] * v2[2];
    cross[2] = v1[0] * v2[1] - v1[1] * v2[0];
}

/* common/mathlib.c:413-429: float length^2, sqrt through double, float reciprocal */
static void vector_normalize(float v[3])
{
    float length, ilength;
    length = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    length = (float)sqrt((double)length);
    if (length) {
        ilength = 1 / length;
        v[0] *= ilength;
        v[1] *= ilength;
        v[2] *= ilength;
    }
}

/* (int)double as x86-64 cvttsd2si does it: out of range / NaN -> INT_MIN
 * (the reference relies on this UB in uv_to_screen / draw_quad) */
static int trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0))
        return INT_MIN;
    return (int)v;
}

/* ---- pure converters (fisheye.c:1184-1214) ----------------------------------- */

void ok_latlon_to_ray(double lat, double lon, float ray[3])
{
    double clat = cos(lat);                 /* fisheye.c:1186 */
    ray[0] = (float)(sin(lon) * clat);      /* :1187  double expr stored to vec_t */
    ray[1] = (float)sin(lat);               /* :1188 */
    ray[2] = (float)(cos(lon) * clat);      /* :1189 */
}

void ok_ray_to_latlon(const float ray[3], double *lat, double *lon)
{
    *lon = atan2(ray[0], ray[2]);                                        /* :1194 */
    /* ray[0]*ray[0]+ray[2]*ray[2] is float arithmetic, sqrt in double     :1195 */
    *lat = atan2(ray[1], sqrt((double)(ray[0] * ray[0] + ray[2] * ray[2])));
}

void ok_plate_uv_to_ray(const ok_state *s, int plate, double u, double v, float ray[3])
{
    const ok_plate *p = &s->plates[plate];
    u -= 0.5;                                /* :1201 */
    v -= 0.5;                                /* :1202 */
    v = -v;                                  /* :1203 */
    ray[0] = ray[1] = ray[2] = 0;            /* :1206 */
    vector_ma(ray, p->dist, p->forward, ray);       /* :1209 */
    vector_ma(ray, (float)u, p->right, ray);        /* :1210  u narrowed to float */
    vector_ma(ray, (float)v, p->up, ray);           /* :1211 */
    vector_normalize(ray);                           /* :1213 */
}

/* ---- what scripts see (fisheye.c:1494-1537) ---------------------------------- */

void ok_lua_latlon_to_ray(double lat, double lon, double out[3])
{
    float ray[3];
    ok_latlon_to_ray(lat, lon, ray);
    out[0] = ray[0]; out[1] = ray[1]; out[2] = ray[2];   /* lua_pushnumber(float) :1500-1502 */
}

void ok_lua_ray_to_latlon(double rx, double ry, double rz, double *lat, double *lon)
{
    float ray[3] = { (float)rx, (float)ry, (float)rz };  /* vec3_t ray = {rx,ry,rz}  :1512 */
    ok_ray_to_latlon(ray, lat, lon);
}

int ok_lua_plate_to_ray(const ok_state *s, double plate, double u, double v, double out[3])
{
    int plate_index = (int)plate;                         /* :1523 */
    float ray[3];
    if (plate_index < 0 || plate_index >= s->numplates)   /* :1527-1530 -> single nil */
        return 0;
    ok_plate_uv_to_ray(s, plate_index, u, v, ray);
    out[0] = ray[0]; out[1] = ray[1]; out[2] = ray[2];
    return 1;
}

static void host_latlon_to_ray(void *ctx, double lat, double lon, double out[3])
{
    (void)ctx;
    ok_lua_latlon_to_ray(lat, lon, out);
}
static void host_ray_to_latlon(void *ctx, double x, double y, double z, double *lat, double *lon)
{
    (void)ctx;
    ok_lua_ray_to_latlon(x, y, z, lat, lon);
}
static int host_plate_to_ray(void *ctx, double plate, double u, double v, double out[3])
{
    return ok_lua_plate_to_ray((const ok_state *)ctx, plate, u, v, out);
}

void ok_default_host(ok_state *s)
{
    s->host.latlon_to_ray = host_latlon_to_ray;
    s->host.ray_to_latlon = host_ray_to_latlon;
    s->host.plate_to_ray = host_plate_to_ray;
    s->host.ctx = s;
}

/* ---- globe loader core (fisheye.c:1796-1869) --------------------------------- */

int ok_set_plate(ok_state *s, int i, const double fwd[3], const double up[3], double fov_deg)
{
    ok_plate *p = &s->plates[i];
    int j;
    for (j = 0; j < 3; ++j) p->forward[j] = (float)fwd[j];     /* :1818 */
    for (j = 0; j < 3; ++j) p->up[j] = (float)up[j];           /* :1843 */
    cross3f(p->up, p->forward, p->right);                      /* :1849 */
    cross3f(p->forward, p->right, p->up);                      /* :1850 (not normalised) */
    p->fov = (float)(fov_deg * M_PI / 180);                    /* :1858 */
    if (p->fov <= 0)                                           /* :1861 */
        return 0;
    p->dist = (float)(0.5 / tan(p->fov / 2));                  /* :1868  fov/2 is float */
    return 1;
}

/* ---- zoom (fisheye.c:1293-1386) ---------------------------------------------- */

int ok_calc_zoom(ok_state *s)
{
    s->scale = -1;                                                        /* :1296 */
    if (s->zoom_type == OK_ZOOM_FOV || s->zoom_type == OK_ZOOM_VFOV) {
        if (s->max_fov <= 0 || s->max_vfov <= 0) return 0;                /* :1301 */
        if (s->zoom_type == OK_ZOOM_FOV && s->zoom_fov > s->max_fov) return 0;   /* :1306 */
        if (s->zoom_type == OK_ZOOM_VFOV && s->zoom_fov > s->max_vfov) return 0; /* :1310 */
        if (s->forward) {                                                 /* :1316 */
            float ray[3];
            double x, y;
            double fovr = s->zoom_fov * M_PI / 180;                       /* :1319 */
            if (s->zoom_type == OK_ZOOM_FOV) {
                ok_latlon_to_ray(0, fovr * 0.5, ray);                     /* :1321 */
                if (s->forward(s->ud, ray[0], ray[1], ray[2], &x, &y) == 1)
                    s->scale = x / (s->width_px * 0.5);                   /* :1323 */
                else
                    return 0;
            } else {
                ok_latlon_to_ray(fovr * 0.5, 0, ray);                     /* :1331 */
                if (s->forward(s->ud, ray[0], ray[1], ray[2], &x, &y) == 1)
                    s->scale = y / (s->height_px * 0.5);                  /* :1333 */
                else
                    return 0;
            }
        } else {
            return 0;                                                     /* :1343 */
        }
    } else if (s->zoom_type == OK_ZOOM_CONTAIN || s->zoom_type == OK_ZOOM_COVER) {
        double fit_width_scale = s->width / s->width_px;                  /* :1349 */
        double fit_height_scale = s->height / s->height_px;               /* :1350 */
        int width_provided = (s->width > 0);
        int height_provided = (s->height > 0);
        if (!width_provided && height_provided) {
            s->scale = fit_height_scale;
        } else if (width_provided && !height_provided) {
            s->scale = fit_width_scale;
        } else if (!width_provided && !height_provided) {
            return 0;
        } else {
            double lens_aspect = s->width / s->height;                    /* :1366 */
            double screen_aspect = (double)s->width_px / s->height_px;    /* :1367 */
            int lens_wider = lens_aspect > screen_aspect;
            if (s->zoom_type == OK_ZOOM_CONTAIN)
                s->scale = lens_wider ? fit_width_scale : fit_height_scale;
            else
                s->scale = lens_wider ? fit_height_scale : fit_width_scale;
        }
    }
    if (s->scale <= 0)                                                    /* :1380 */
        return 0;
    return 1;
}

/* ---- lens pixel setters (fisheye.c:1922-2013) -------------------------------- */

static void set_lensmap_grid(ok_state *s, int lx, int ly, int px, int py, int plate)
{
    double block_size = (s->rubix_pad + s->rubix_cell);                   /* :1938 */
    double num_units = s->rubix_numcells * block_size + s->rubix_pad;     /* :1945 */
    double unit_size_px = (double)s->platesize / num_units;               /* :1948 */
    double ux = (double)px / unit_size_px;                                /* :1951 */
    double uy = (double)py / unit_size_px;                                /* :1952 */
    int ongrid = fmod(ux, block_size) < s->rubix_pad ||
                 fmod(uy, block_size) < s->rubix_pad;                     /* :1954-1956 */
    if (!ongrid)
        s->tints[lx + ly * s->width_px] = (uint8_t)plate;                 /* :1959 */
}

static void set_lensmap_from_plate(ok_state *s, int lx, int ly, int px, int py, int plate)
{
    if (lx < 0 || lx >= s->width_px || ly < 0 || ly >= s->height_px)      /* :1966 */
        return;
    if (px < 0 || px >= s->platesize || py < 0 || py >= s->platesize)     /* :1971 */
        return;
    s->plates[plate].display = 1;                                         /* :1976 */
    s->offsets[lx + ly * s->width_px] =                                   /* :1979 / GLOBEPIXEL :349 */
        (uint32_t)plate * (uint32_t)s->platesize * (uint32_t)s->platesize +
        (uint32_t)px + (uint32_t)py * (uint32_t)s->platesize;
    set_lensmap_grid(s, lx, ly, px, py, plate);                           /* :1981 */
}

static void set_lensmap_from_plate_uv(ok_state *s, int lx, int ly, double u, double v, int plate)
{
    int px = trunc_to_int(u * s->platesize);                              /* :1988 */
    int py = trunc_to_int(v * s->platesize);                              /* :1989 */
    set_lensmap_from_plate(s, lx, ly, px, py, plate);
}

/* fisheye.c:2023-2050 */
static int ray_to_plate_index(const ok_state *s, const float ray[3])
{
    int plate_index = 0;
    double max_dp = -2;
    int i;
    if (s->globe_plate) {                                                 /* :2027-2033 */
        if (s->globe_plate(s->ud, ray[0], ray[1], ray[2], &plate_index))
            return plate_index;
        return -1;
    }
    for (i = 0; i < s->numplates; ++i) {
        double dp = dot3f(ray, s->plates[i].forward);                     /* :2042 float dot */
        if (dp > max_dp) {                                                /* strict >: first max wins */
            max_dp = dp;
            plate_index = i;
        }
    }
    return plate_index;
}

/* fisheye.c:2052-2066 */
static int ray_to_plate_uv(const ok_state *s, int plate, const float ray[3], double *u, double *v)
{
    const ok_plate *p = &s->plates[plate];
    double x = dot3f(p->right, ray);
    double y = dot3f(p->up, ray);
    double z = dot3f(p->forward, ray);
    double dist = 0.5 / tan(p->fov / 2);          /* :2060 recomputed in double from float fov */
    *u = x / z * dist + 0.5;                      /* :2061 */
    *v = -y / z * dist + 0.5;                     /* :2062 */
    return *u >= 0 && *u <= 1 && *v >= 0 && *v <= 1;
}

/* fisheye.c:1995-2013 */
static void set_lensmap_from_ray(ok_state *s, int lx, int ly, double sx, double sy, double sz)
{
    float ray[3] = { (float)sx, (float)sy, (float)sz };
    double u, v;
    int plate = ray_to_plate_index(s, ray);
    if (plate < 0)
        return;
    /* NB: an out-of-range index from a globe_plate override is not guarded in
     * the reference either (fisheye.c:2029-2031); scripts return valid ones. */
    if (!ray_to_plate_uv(s, plate, ray, &u, &v))
        return;
    set_lensmap_from_plate_uv(s, lx, ly, u, v, plate);
}

/* ---- inverse build (fisheye.c:2084-2124, 1545-1588) -------------------------- */

int ok_build_inverse_rows(ok_state *s, int y0, int y1)
{
    int lx, ly;
    for (ly = y1 - 1; ly >= y0; --ly) {                                   /* rows bottom-up */
        double y = -(ly - s->height_px / 2) * s->scale;                   /* :2100 integer H/2 */
        for (lx = 0; lx < s->width_px; ++lx) {
            double x = (lx - s->width_px / 2) * s->scale;                 /* :2105 */
            double r[3];
            float ray[3];
            int status = s->inverse(s->ud, x, y, r);                      /* :2109 */
            if (status == 0) continue;                                    /* :2110 */
            if (status == -1) return 0;                                   /* :2113 */
            ray[0] = (float)r[0]; ray[1] = (float)r[1]; ray[2] = (float)r[2];   /* :1559-1561 */
            vector_normalize(ray);                                        /* :1562 */
            set_lensmap_from_ray(s, lx, ly, ray[0], ray[1], ray[2]);      /* :2118 */
        }
    }
    return 1;
}

/* ---- forward build (fisheye.c:2126-2338), run to completion ------------------ */

/* fisheye.c:2227-2243.  Returns 1 / 0 (nil) / -1 */
static int uv_to_screen(ok_state *s, int plate, double u, double v, int *lx, int *ly)
{
    float ray[3];
    double x, y;
    int status;
    ok_plate_uv_to_ray(s, plate, u, v, ray);
    status = s->forward(s->ud, ray[0], ray[1], ray[2], &x, &y);
    if (status == 0 || status == -1) return status;
    *lx = trunc_to_int(x / s->scale + s->width_px / 2);                   /* :2239 integer W/2 */
    *ly = trunc_to_int(-y / s->scale + s->height_px / 2);                 /* :2240 */
    return status;
}

/* fisheye.c:2246-2338 (compiled -fwrapv: the reference's int overflow on
 * INT_MIN corner coordinates wraps on x86-64) */
static void draw_quad(ok_state *s, const int *tl, const int *tr, const int *bl, const int *br,
                      int plate, int px, int py)
{
    const int *p[4];
    int x = tl[0], y = tl[1];
    int miny = y, maxy = y, minx = x, maxx = x;
    int i;
    const int maxdiff = 20;
    p[0] = tl; p[1] = tr; p[2] = br; p[3] = bl;                           /* clockwise :2250 */
    for (i = 1; i < 4; i++) {
        int tx = p[i][0], ty = p[i][1];
        if (tx < minx) minx = tx; else if (tx > maxx) maxx = tx;          /* :2259-2260 */
        if (ty < miny) miny = ty; else if (ty > maxy) maxy = ty;          /* :2263-2264 */
    }
    if (abs(minx - maxx) > maxdiff || abs(miny - maxy) > maxdiff)         /* :2272 */
        return;
    if (miny == maxy && minx == maxx) {                                   /* :2277 */
        set_lensmap_from_plate(s, x, y, px, py, plate);
        return;
    }
    if (miny == maxy) {                                                   /* :2283 */
        int tx;
        for (tx = minx; tx <= maxx; ++tx)
            set_lensmap_from_plate(s, tx, miny, px, py, plate);
        return;
    }
    if (minx == maxx) {                                                   /* :2292 */
        int ty;
        for (ty = miny; ty <= maxy; ++ty)
            set_lensmap_from_plate(s, x, ty, px, py, plate);
        return;
    }
    for (y = miny; y <= maxy; ++y) {                                      /* :2301 */
        int tx[2];
        int txi = 0, j = 3;
        tx[0] = minx; tx[1] = maxx;
        for (i = 0; i < 4; ++i) {
            int ix = p[i][0], iy = p[i][1];
            int jx = p[j][0], jy = p[j][1];
            if ((iy < y && y <= jy) || (jy < y && y <= iy)) {             /* :2310 */
                double dy = jy - iy;
                double dx = jx - ix;
                tx[txi] = trunc_to_int(ix + (y - iy) / dy * dx);          /* :2313 */
                if (++txi == 2) break;
            }
            j = i;
        }
        if (tx[0] > tx[1]) { int t = tx[0]; tx[0] = tx[1]; tx[1] = t; }   /* :2320 */
        if (tx[1] - tx[0] > maxdiff)                                      /* :2327 aborts the quad */
            return;
        for (x = tx[0]; x <= tx[1]; ++x)
            set_lensmap_from_plate(s, x, y, px, py, plate);
    }
}

/* TEST HOOK: draw_quad alone on an empty W x H table (one 8 x 8 plate, texel (0,0)): mask[y * W + x] = 1 where it wrote.
 * corners = tl.x, tl.y, tr.x, tr.y, bl.x, bl.y, br.x, br.y (tests/test_exactness_cpu.py pins the INT_MIN cases with it). */
void ok_test_draw_quad(int W, int H, const int corners[8], unsigned char *mask)
{
    ok_state s;
    size_t area = (size_t)W * (size_t)H, i;
    memset(&s, 0, sizeof s);
    s.width_px = W; s.height_px = H; s.platesize = 8; s.numplates = 1;
    s.rubix_numcells = 10; s.rubix_cell = 4; s.rubix_pad = 1;
    s.offsets = (uint32_t *)malloc(area * sizeof(uint32_t));
    s.tints = (uint8_t *)malloc(area);
    for (i = 0; i < area; ++i) s.offsets[i] = OK_NULL_OFFSET;
    memset(s.tints, 255, area);
    draw_quad(&s, corners + 0, corners + 2, corners + 4, corners + 6, 0, 0, 0);
    for (i = 0; i < area; ++i) mask[i] = s.offsets[i] != OK_NULL_OFFSET;
    free(s.offsets); free(s.tints);
}

/* fisheye.c:2126-2217 with seconds_per_frame = infinity.
 * Defined behaviour for a nil corner (the reference reads a stale / uninitialised
 * entry there; no shipped forward lens returns nil): the quads touching that
 * corner are skipped. */
static int build_forward(ok_state *s)
{
    int ps = s->platesize;
    int *rowa = (int *)malloc((size_t)(ps + 1) * 2 * sizeof(int));
    int *rowb = (int *)malloc((size_t)(ps + 1) * 2 * sizeof(int));
    unsigned char *vala = (unsigned char *)malloc((size_t)ps + 1);
    unsigned char *valb = (unsigned char *)malloc((size_t)ps + 1);
    int *top = rowa, *bot = rowb;
    unsigned char *topv = vala, *botv = valb;
    int plate, py, px, ok = 1;

    for (plate = 0; plate < s->numplates && ok; ++plate) {
        for (py = ps - 1; py >= 0 && ok; --py) {
            if (py == ps - 1) {                                           /* :2148 lower points */
                double v = (py + 0.5) / ps;
                for (px = 0; px < ps; ++px) {
                    int st;
                    if (px == 0) {
                        double u = (px - 0.5) / ps;
                        st = uv_to_screen(s, plate, u, v, &bot[0], &bot[1]);
                        if (st == -1) { ok = 0; break; }
                        botv[0] = (unsigned char)(st == 1);
                    }
                    {
                        double u = (px + 0.5) / ps;
                        int index = 2 * (px + 1);
                        st = uv_to_screen(s, plate, u, v, &bot[index], &bot[index + 1]);
                        if (st == -1) { ok = 0; break; }
                        botv[px + 1] = (unsigned char)(st == 1);
                    }
                }
                if (!ok) break;
            } else {                                                      /* :2164 swap rows */
                int *t = top; unsigned char *tv = topv;
                top = bot; bot = t;
                topv = botv; botv = tv;
            }
            {                                                             /* :2172 upper points */
                double v = (py - 0.5) / ps;
                for (px = 0; px < ps; ++px) {
                    int st;
                    if (px == 0) {
                        double u = (px - 0.5) / ps;
                        st = uv_to_screen(s, plate, u, v, &top[0], &top[1]);
                        if (st == -1) { ok = 0; break; }
                        topv[0] = (unsigned char)(st == 1);
                    }
                    {
                        double u = (px + 0.5) / ps;
                        int index = 2 * (px + 1);
                        st = uv_to_screen(s, plate, u, v, &top[index], &top[index + 1]);
                        if (st == -1) { ok = 0; break; }
                        topv[px + 1] = (unsigned char)(st == 1);
                    }
                }
                if (!ok) break;
            }
            {                                                             /* :2189 draw quads */
                double v = ((double)py) / ps;
                for (px = 0; px < ps; ++px) {
                    double u = ((double)px) / ps;
                    float ray[3];
                    int index = 2 * px;
                    ok_plate_uv_to_ray(s, plate, u, v, ray);
                    if (plate != ray_to_plate_index(s, ray))              /* :2196 */
                        continue;
                    if (!(topv[px] && topv[px + 1] && botv[px] && botv[px + 1]))
                        continue;
                    draw_quad(s, &top[index], &top[index + 2], &bot[index], &bot[index + 2],
                              plate, px, py);
                }
            }
        }
    }
    free(rowa); free(rowb); free(vala); free(valb);
    return ok;
}

/* ---- create_lensmap (fisheye.c:2367-2397) + the clears of F_RenderView:731-732 */

int ok_create_lensmap(ok_state *s)
{
    size_t area = (size_t)s->width_px * (size_t)s->height_px;
    size_t i;
    int k;
    for (i = 0; i < area; ++i) s->offsets[i] = OK_NULL_OFFSET;            /* memset NULL :731 */
    memset(s->tints, 255, area);                                          /* :732 */
    if (!ok_calc_zoom(s))                                                 /* :2376 */
        return 0;
    for (k = 0; k < s->numplates; k++) s->plates[k].display = 0;          /* :2383 */
    if (s->map_type == OK_MAP_FORWARD)
        return build_forward(s);
    if (s->map_type == OK_MAP_INVERSE)
        return ok_build_inverse_rows(s, 0, s->height_px);
    return 0;
}

/* ---- apply (fisheye.c:2406-2424) --------------------------------------------- */

void ok_apply(const ok_state *s, const uint8_t *globe, uint8_t *dst, int dst_pitch,
              int x0, int y0, int rubix_on)
{
    const uint32_t *lmap = s->offsets;
    const uint8_t *pmap = s->tints;
    int x, y;
    for (y = 0; y < s->height_px; y++)
        for (x = 0; x < s->width_px; x++, lmap++, pmap++)
            if (*lmap != OK_NULL_OFFSET) {
                uint8_t *out = dst + (x + x0) + (size_t)(y + y0) * dst_pitch;   /* VBUFFER :634 */
                if (rubix_on) {
                    int i = *pmap;
                    *out = i != 255 ? s->plates[i].palette[globe[*lmap]] : globe[*lmap];  /* :2418 */
                } else {
                    *out = globe[*lmap];                                  /* :2421 */
                }
            }
}

/* ---- rubix palettes (fisheye.c:835-908) -------------------------------------- */

static int find_closest_pal_index(const uint8_t *basepal, int r, int g, int b)
{
    int i, mindist = 256 * 256 * 256, minindex = 0;
    const uint8_t *pal = basepal;
    for (i = 0; i < 256; ++i) {
        int dr = (int)pal[0] - r, dg = (int)pal[1] - g, db = (int)pal[2] - b;
        int dist = dr * dr + dg * dg + db * db;
        if (dist < mindist) { mindist = dist; minindex = i; }            /* first min wins */
        pal += 3;
    }
    return minindex;
}

void ok_create_palmap(ok_state *s, const uint8_t *basepal)
{
    static const int tints[OK_MAX_PLATES][3] = {                          /* :866-886 */
        {255, 255, 255}, {0, 0, 255}, {255, 0, 0}, {255, 255, 0}, {255, 0, 255}, {0, 255, 255}
    };
    int i, j, percent = 256 / 6;                                          /* :860 */
    for (j = 0; j < OK_MAX_PLATES; ++j) {
        const uint8_t *pal = basepal;
        for (i = 0; i < 256; ++i) {
            int r = pal[0], g = pal[1], b = pal[2];
            r += percent * (tints[j][0] - r) >> 8;                        /* :895-897 */
            g += percent * (tints[j][1] - g) >> 8;
            b += percent * (tints[j][2] - b) >> 8;
            if (r < 0) r = 0; if (r > 255) r = 255;
            if (g < 0) g = 0; if (g > 255) g = 255;
            if (b < 0) b = 0; if (b > 255) b = 255;
            s->plates[j].palette[i] = (uint8_t)find_closest_pal_index(basepal, r, g, b);
            pal += 3;
        }
    }
}

/* ---- test helpers ------------------------------------------------------------ */

/* WritePCXplate, fisheye.c:1396-1465 (itself "copied from WritePCXfile in NQ/screen.c"); pcx_t is
 * NQ/client.h:376-391: 128 header bytes, then the data.  Little-endian shorts (LittleShort). */
int ok_write_pcx_plate(const ok_state *s, int plate, int with_margins, const uint8_t *plate_pixels,
                       const uint8_t *basepal, uint8_t *out)
{
    const int width = s->platesize, height = s->platesize;                /* :1400-1405 */
    const uint8_t *data = plate_pixels;
    uint8_t *pack;
    int i, j;
    memset(out, 0, 128);
    out[0] = 0x0a;                                  /* manufacturer: PCX id            :1419 */
    out[1] = 5;                                     /* version: 256 color              :1420 */
    out[2] = 1;                                     /* encoding                        :1421 */
    out[3] = 8;                                     /* bits_per_pixel                  :1422 */
    /* xmin, ymin = 0 */
    out[8] = (uint8_t)((width - 1) & 0xFF);  out[9] = (uint8_t)(((width - 1) >> 8) & 0xFF);     /* xmax  :1425 */
    out[10] = (uint8_t)((height - 1) & 0xFF); out[11] = (uint8_t)(((height - 1) >> 8) & 0xFF);  /* ymax  :1426 */
    out[12] = (uint8_t)(width & 0xFF);  out[13] = (uint8_t)((width >> 8) & 0xFF);               /* hres  :1427 */
    out[14] = (uint8_t)(height & 0xFF); out[15] = (uint8_t)((height >> 8) & 0xFF);              /* vres  :1428 */
    /* palette[48] = 0, reserved = 0 (Hunk_TempAlloc memory; the reference leaves `reserved` unset) */
    out[65] = 1;                                    /* color_planes: chunky image      :1430 */
    out[66] = (uint8_t)(width & 0xFF); out[67] = (uint8_t)((width >> 8) & 0xFF);                /* bytes_per_line :1431 */
    out[68] = 2; out[69] = 0;                       /* palette_type: not a grey scale  :1432 */
    pack = out + 128;                               /* &pcx->data                      :1436 */
    for (i = 0; i < height; i++) {
        double v = ((double)i) / height;                                    /* :1439 */
        for (j = 0; j < width; j++) {
            double u = ((double)j) / width;                                 /* :1441 */
            float ray[3];
            uint8_t col;
            ok_plate_uv_to_ray(s, plate, u, v, ray);
            col = (with_margins || plate == ray_to_plate_index(s, ray)) ? *data : 0xFE;     /* :1446 */
            if ((col & 0xc0) == 0xc0) *pack++ = 0xc1;                       /* :1448-1450 */
            *pack++ = col;
            data++;
        }
    }
    *pack++ = 0x0c;                                 /* palette ID byte                 :1459 */
    for (i = 0; i < 768; i++) *pack++ = basepal[i];
    return (int)(pack - out);
}

uint64_t ok_fnv1a64(const void *data, size_t n)
{
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    size_t i;
    for (i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* s0 = 0x9E3779B9*(p+1+6*frame); s = s*1664525+1013904223; texel = s>>24 */
void ok_lcg_fill_plate(uint8_t *dst, size_t n, int plate, int frame)
{
    uint32_t st = 0x9E3779B9u * (uint32_t)(plate + 1 + 6 * frame);
    size_t i;
    for (i = 0; i < n; ++i) {
        st = st * 1664525u + 1013904223u;
        dst[i] = (uint8_t)(st >> 24);
    }
}
