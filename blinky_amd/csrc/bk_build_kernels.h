/* bk_build_kernels.h -- the generic lensmap BUILD kernels (embedded last into the hiprtc unit).
 * The emitter provides LF_lens_inverse / LF_lens_forward / LF_globe_plate (when the scripts
 * define them), BK_HAS_* and BK_INIT_GLOBALS.  One thread = one output pixel (inverse map) or
 * one plate texel / texel corner (forward map).  Arithmetic follows fisheye.c line by line:
 * float where the reference uses vec_t, double elsewhere, no contraction. */

BK_DEV void bk_state_init(BkState &S, const BkBuildParams *P)
{
    S.P = P;
    S.err = 0;
    S.steps = 0;
    S.flag = 0;
    BK_INIT_GLOBALS(S)
}

/* fisheye.c:2023-2050 */
BK_DEV int bk_ray_to_plate_index(BkState &S, const float *ray)
{
    const BkBuildParams &P = *S.P;
#ifdef BK_HAS_GLOBE_PLATE
    bkv a[3] = {bk_num((double)ray[0]), bk_num((double)ray[1]), bk_num((double)ray[2])};
    bkv r[BK_MAXRET];
    int n = LF_globe_plate(S, a, 3, r);
    if (n < 1 || !bk_isnum(r[n - 1])) return -1;            /* !lua_isnumber(lua,-1)  :1642 */
    double d = r[n - 1].n;                                  /* lua_tointeger: round to nearest (LUA_IEEE754TRICK) */
    if (r[n - 1].e != 0.0 && !(bkm_rint(d - r[n - 1].e) == bkm_rint(d + r[n - 1].e))) S.flag = 1;
    if (!(d > -2147483648.0 && d < 2147483647.0)) return -1;
    int plate = (int)bkm_rint(d);
    if (plate < 0 || plate >= P.numplates) return -1;       /* the reference would index out of bounds */
    return plate;
#else
    /* (the reference widens each float dot to double and compares those, :2042: widening is exact and keeps the order, NaN included -
     * the floats compare the same.  Written out for the six plates a globe can have: the plate vectors arrive in one scalar load
     * instead of one per turn of a loop.) */
    int plate_index = 0;
    float max_dp = -2.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (i < P.numplates) {
            const float dp = bk_dot3(ray, P.plates[i].forward);
            if (dp > max_dp) { max_dp = dp; plate_index = i; }   /* strict >: first maximum wins */
        }
    }
    return plate_index;
#endif
}

/* set_lensmap_grid, fisheye.c:1922-1960: true when the texel is NOT on a grid line */
BK_DEV bool bk_offgrid(const BkBuildParams &P, int px, int py)
{
    if (P.grid_n)                                     /* (0 <= px, py < ps <= grid_n: the callers' range checks) */
        return !(((P.grid_bits[px >> 5] >> (px & 31)) | (P.grid_bits[py >> 5] >> (py & 31))) & 1u);
    double ux = (double)px / P.rubix_unit_px;
    double uy = (double)py / P.rubix_unit_px;
    bool ongrid = bkm_fmod(ux, P.rubix_block) < P.rubix_pad || bkm_fmod(uy, P.rubix_block) < P.rubix_pad;
    return !ongrid;
}

BK_DEV unsigned int bk_padded_offset(const BkBuildParams &P, int plate, int px, int py)
{
    return bk_texel_offset((unsigned int)P.gp, (unsigned int)P.ph, (unsigned int)plate, (unsigned int)px, (unsigned int)py);
}

#ifndef BK_HOST_MODULE
/* a pixel / corner / texel whose outcome the host has to re-derive on the platform libm */
__device__ __forceinline__ void bk_push_flagged(const BkBuildParams &P, unsigned int id, unsigned int a, unsigned int b, unsigned int c)
{
    const unsigned int k = atomicAdd(P.flag_count, 1u);
    if (k < P.flag_cap) {
        P.flag_list[4 * k] = id; P.flag_list[4 * k + 1] = a; P.flag_list[4 * k + 2] = b; P.flag_list[4 * k + 3] = c;
    }
}

/* The same for a whole wave at once (every lane that is still active calls it): the flagged lanes take CONSECUTIVE slots, in lane
 * order - one atomic per wave instead of one per entry, and the list arrives in runs of ascending ids (a wave covers 64 consecutive
 * pixels of a row / 64 consecutive corners), which is what keeps a script's per-row caches warm when the host walks it (eckert4). */
__device__ __forceinline__ void bk_push_flagged_wave(const BkBuildParams &P, bool flag, unsigned int id, unsigned int a, unsigned int b, unsigned int c)
{
    const unsigned long long m = __ballot(flag);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(P.flag_count, (unsigned int)__popcll(m));
    base = (unsigned int)__shfl((int)base, leader);
    if (flag) {
        const unsigned int k = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (k < P.flag_cap) {
            P.flag_list[4 * k] = id; P.flag_list[4 * k + 1] = a; P.flag_list[4 * k + 2] = b; P.flag_list[4 * k + 3] = c;
        }
    }
}

__device__ __forceinline__ void bk_publish_flags(const int *s_disp, int *display, int serr, int *err)
{
    /* per-block reduction of the display[] flags: at most 6 atomics per block, none once set */
    if (threadIdx.x < 6 && s_disp[threadIdx.x]) {
        if (__hip_atomic_load(&display[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            atomicOr(&display[threadIdx.x], 1);
    }
    if (serr) atomicOr(err, serr);
}
#endif

#ifdef BK_HAS_INVERSE
/* one pixel of resume_lensmap_inverse + LUAtoC_lens_inverse + set_lensmap_from_ray (fisheye.c:2084-2124, 1545-1588, 1995-2013):
 * the lensmap entry of screen pixel (lx, ly) and the plate it shows (-1: none).  Shared by the build kernel and - compiled for
 * the host against the platform libm - by the re-derivation of the flagged pixels (bk_hostmod_driver.inc). */
BK_DEV void bk_inverse_entry(const BkBuildParams &P, BkState &S, int lx, int ly, unsigned int *off_out, unsigned char *tint_out, int *plate_out)
{
    unsigned int off = 0xFFFFFFFFu;
    unsigned char tint = 255;
    int shown = -1;
    const double y = (double)(-(ly - P.H / 2)) * P.scale;     /* :2100, integer H/2 */
    const double x = (double)(lx - P.W / 2) * P.scale;        /* :2105 */
    bkv a[2] = {bk_num(x), bk_num(y)};
    bkv r[BK_MAXRET];
    const int n = LF_lens_inverse(S, a, 2, r);
    if (n == 3 && bk_isnum(r[0]) && bk_isnum(r[1]) && bk_isnum(r[2])) {
        float ray[3] = {bk_narrow(S, r[0].n, r[0].e), bk_narrow(S, r[1].n, r[1].e), bk_narrow(S, r[2].n, r[2].e)};   /* :1559-1561 */
        bk_vector_normalize(ray);                                        /* :1562 */
        const int plate = bk_ray_to_plate_index(S, ray);
        /* from the float ray on, every step is an IEEE operation the reference performs identically */
        if (plate >= 0) {
            const BkPlateDev &p = P.plates[plate];
            const double px_ = (double)bk_dot3(p.right, ray);            /* :2055-2057 */
            const double py_ = (double)bk_dot3(p.up, ray);
            const double pz_ = (double)bk_dot3(p.forward, ray);
            const double u = px_ / pz_ * p.dist64 + 0.5;                 /* :2061 */
            const double v = -py_ / pz_ * p.dist64 + 0.5;                /* :2062 */
            if (u >= 0 && u <= 1 && v >= 0 && v <= 1) {                  /* :2065 */
                const int px = bk_trunc_to_int(u * P.ps);                /* :1988 */
                const int py = bk_trunc_to_int(v * P.ps);
                if (px >= 0 && px < P.ps && py >= 0 && py < P.ps) {      /* :1971 */
                    shown = plate;                                       /* :1976 */
                    off = bk_padded_offset(P, plate, px, py);            /* :1979 */
                    if (bk_offgrid(P, px, py)) tint = (unsigned char)plate;   /* :1959 */
                }
            }
        }
    } else if (!(n == 1 && r[0].t == BK_TNIL)) {
        S.err |= BK_ERR_RESULT;                                          /* status -1  :1565-1584 */
    }
    *off_out = off;
    *tint_out = tint;
    *plate_out = shown;
}

#ifndef BK_HOST_MODULE
extern "C" __global__ __launch_bounds__(256) void bk_build_inverse(BkBuildParams P)
{
    __shared__ int s_disp[6];
    if (threadIdx.x < 6) s_disp[threadIdx.x] = 0;
    __syncthreads();
    const int lx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lyl = blockIdx.y;                      /* row inside the owned stripe */
    const int ly = P.row0 + lyl;
    int err = 0;
    bool flagged = false;
    unsigned int off = 0xFFFFFFFFu;
    unsigned char tint = 255;
    const size_t o = (size_t)lyl * P.W + lx;
    if (lx < P.W) {
        int plate;
        BkState S;
        bk_state_init(S, &P);
        bk_inverse_entry(P, S, lx, ly, &off, &tint, &plate);
        if (plate >= 0 && !S.flag) s_disp[plate] = 1;    /* (a flagged pixel's plate is the host's to say) */
        err = S.err;
        P.offsets[o] = off;
        P.tints[o] = tint;
        flagged = S.flag != 0;
        /* status -1 (fisheye.c:2113-2115) ends the reference's scan there: remember the first such pixel in ITS order */
        if ((err & BK_ERR_RESULT) && P.first_bad) atomicMax(P.first_bad, (unsigned int)(ly * P.W + (P.W - 1 - lx)) + 1u);
    }
    bk_push_flagged_wave(P, flagged, (unsigned int)o, off, tint, 0u);
    __syncthreads();
    bk_publish_flags(s_disp, P.display, err, P.err);
}
#endif
#endif

#ifdef BK_HAS_FORWARD
/* (int)(v / scale + half), uv_to_screen's last step (fisheye.c:2236-2240), and the question whether another libm's v (within e) would give
 * another pixel.  On the device nearly every corner is answered without the division: fa = v * (1 / scale) + half is within
 * 2^-49 * (|v| / scale + half) of the reference's rounded f, so when trunc is the same number T over [fa - ea, fa + ea] - ea that, doubled
 * twice over, plus the libm bound - T is trunc(f) and every admissible f's trunc: no flag, no division (and T in int range is what the
 * x86 conversion gives).  The few corners within 1e-11 of a pixel's edge take the reference's own operations, as before. */
BK_DEV int bk_to_screen(const BkBuildParams &P, BkState &S, double v, double e, int half)
{
#ifndef BK_HOST_MODULE
    {
        const double fa = __builtin_fma(v, P.inv_scale_up, (double)half);
        const double ea = __builtin_fma(__builtin_fma(bk_abs(v), P.inv_scale_up, (double)half), 0x1p-47, e * P.inv_scale_up);
        const double lo = bkm_trunc(fa - ea);
        if (lo == bkm_trunc(fa + ea) && lo > -2147483649.0 && lo < 2147483648.0) return (int)lo;
    }
#endif
    const double f = v / P.scale + (double)half;
    bk_need_same_trunc(S, f, bk_eop(S, f, e * P.inv_scale_up));
    return bk_trunc_to_int(f);                                               /* :2239-2240 */
}

/* uv_to_screen (fisheye.c:2227-2243) for every texel-corner of every plate:
 * corner (i,j), i,j in 0..ps, is (u,v) = ((i-0.5)/ps, (j-0.5)/ps) */
BK_DEV void bk_corner_at(const BkBuildParams &P, BkState &S, int plate, int j, int i, int *sx_out, int *sy_out, unsigned char *ok_out)
{
    float ray[3];
#ifdef BK_HOST_MODULE
    const double u = ((double)i - 0.5) / P.ps;
    const double v = ((double)j - 0.5) / P.ps;
    bk_plate_uv_to_ray(P, plate, u, v, ray);
#else
    bk_plate_fuv_to_ray(P, plate, P.fwd_uv[i], -P.fwd_uv[j], ray);          /* the same two floats, from the build's table */
#endif
    bkv a[3] = {bk_num((double)ray[0]), bk_num((double)ray[1]), bk_num((double)ray[2])};
    bkv r[BK_MAXRET];
    const int n = LF_lens_forward(S, a, 3, r);
    unsigned char ok = 0;
    int sx = 0, sy = 0;
    if (n == 2 && bk_isnum(r[0]) && bk_isnum(r[1])) {
        sx = bk_to_screen(P, S, r[0].n, r[0].e, P.W / 2);                    /* r[0].n / scale + W / 2, :2236 */
        sy = bk_to_screen(P, S, -r[1].n, r[1].e, P.H / 2);                   /* -r[1].n / scale + H / 2, :2237 */
        ok = 1;
    } else if (!(n == 1 && r[0].t == BK_TNIL)) {
        S.err |= BK_ERR_RESULT;
    }
    *sx_out = sx;
    *sy_out = sy;
    *ok_out = ok;
}
/* the same by corner number id = (plate * (ps+1) + j) * (ps+1) + i (the flagged list's and the host module's way to name one) */
BK_DEV void bk_corner_entry(const BkBuildParams &P, BkState &S, long long id, int *sx_out, int *sy_out, unsigned char *ok_out)
{
    const int n1 = P.ps + 1;
    const int plate = (int)(id / ((long long)n1 * n1));
    const int rem = (int)(id - (long long)plate * n1 * n1);
    const int j = rem / n1, i = rem - j * n1;
    bk_corner_at(P, S, plate, j, i, sx_out, sy_out, ok_out);
}

/* forward build: does the ray through texel `id` select its own plate (fisheye.c:2193-2196) */
BK_DEV bool bk_texel_owns_at(const BkBuildParams &P, BkState &S, int plate, int py, int px)
{
    float ray[3];
#ifdef BK_HOST_MODULE
    bk_plate_uv_to_ray(P, plate, (double)px / P.ps, (double)py / P.ps, ray);   /* :2193-2195 */
#else
    bk_plate_fuv_to_ray(P, plate, P.fwd_uv[P.ps + 1 + px], -P.fwd_uv[P.ps + 1 + py], ray);
#endif
    return plate == bk_ray_to_plate_index(S, ray);                              /* :2196 */
}
BK_DEV bool bk_texel_owns(const BkBuildParams &P, BkState &S, long long id)     /* id = (plate * ps + py) * ps + px */
{
    const int plate = (int)(id / ((long long)P.ps * P.ps));
    const int rem = (int)(id - (long long)plate * P.ps * P.ps);
    const int py = rem / P.ps, px = rem - py * P.ps;
    return bk_texel_owns_at(P, S, plate, py, px);
}

#endif

/* ---- everything below is device-only (the host module of the flagged-entry re-derivation stops here) ---- */
#ifndef BK_HOST_MODULE
#ifdef BK_HAS_FORWARD
/* grid (ceil((ps+1)/256), ps+1, plates): a corner's plate and row come from the block, nothing is divided */
extern "C" __global__ __launch_bounds__(256) void bk_forward_corners(BkBuildParams P)
{
    const int n1 = P.ps + 1;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), j = (int)blockIdx.y, plate = (int)blockIdx.z;
    const bool live = i < n1;
    const unsigned int id = ((unsigned int)plate * (unsigned int)n1 + (unsigned int)j) * (unsigned int)n1 + (unsigned int)i;
    BkState S;
    bk_state_init(S, &P);
    unsigned char ok = 0;
    int sx = 0, sy = 0;
    if (live) {
        bk_corner_at(P, S, plate, j, i, &sx, &sy, &ok);
        P.corner_xy[2 * (size_t)id] = sx;
        P.corner_xy[2 * (size_t)id + 1] = sy;
        P.corner_ok[id] = ok;
    }
    bk_push_flagged_wave(P, live && S.flag != 0, id, (unsigned int)sx, (unsigned int)sy, ok);
    if (live && S.err) atomicOr(P.err, S.err);
}

/* set_lensmap_from_plate (fisheye.c:1963-1982) as an ordered-overwrite commit */
/* A workgroup's window on the screen, in LDS: a tile of neighbouring texels lands on a small patch of pixels, several texels per
 * pixel under a minifying lens (28 M texels on 8 M pixels at 4K, most of them straddling two to four) - 64 M atomicMax to memory,
 * which is what the quad pass cost (2.7 ms of the 3.4 ms of a 4K forward build; profiles/r04_build_counters.txt).  The patch is
 * reduced in LDS first and written out once; what falls outside the window goes to memory directly, as before. */
#define BK_FWD_TILE 16
#define BK_FWD_WIN 48
struct BkFwdWin {
    int x0, y0, w, h;                 /* window on the screen: origin, extent (<= BK_FWD_WIN each; rows are BK_FWD_WIN apart in LDS) */
    unsigned int *px, *tint;          /* [BK_FWD_WIN * BK_FWD_WIN] keys, 0 = none; px == nullptr: no window */
    const double *quot;               /* BkBuildParams::fwd_quot */
};
/* (double)a / (double)d of draw_quad's edge interpolation (fisheye.c:2313).  An edge of a quad that passed the 20-pixel size check has
 * |d| <= 20 and a between 0 and d: the quotient is |a| / |d| - IEEE division is sign-symmetric - and comes from the build's table
 * (a = 0 over a negative d reads +0 where the division gives -0: times dx, plus an integer, truncated, the same int).  Anything
 * else - the wrapped differences of INT_MIN corners - is divided out as before. */
BK_DEV double bk_edge_quot(int a, int d, const double *quot)
{
    const unsigned int ua = a < 0 ? 0u - (unsigned int)a : (unsigned int)a, ud = d < 0 ? 0u - (unsigned int)d : (unsigned int)d;
    if (ua <= 20u && ud - 1u < 20u && ((a ^ d) >= 0 || a == 0)) return quot[ua * 21u + ud];
    return (double)a / (double)d;
}
/* set_lensmap_from_plate (fisheye.c:1963-1982) for the pixels [xa, xb] of screen row ly: the row's tests made once */
BK_DEV void bk_fwd_row(const BkBuildParams &P, int xa, int xb, int ly, unsigned int key, bool offgrid, int *wrote, const BkFwdWin &win)
{
    if (ly < 0 || ly >= P.H) return;                                         /* :1966 */
    xa = xa < 0 ? 0 : xa;
    xb = xb >= P.W ? P.W - 1 : xb;
    if (xa > xb) return;
    *wrote = 1;                                                               /* display (:1976) is global, not per stripe */
    if (ly < P.row0 || ly >= P.row0 + P.rows) return;                        /* stripe-filtered commit */
    const int wy = ly - win.y0;
    const bool row_in = win.px && wy >= 0 && wy < win.h;
    for (int lx = xa; lx <= xb; ++lx) {
        const int wx = lx - win.x0;
        if (row_in && wx >= 0 && wx < win.w) {
            const int k = wy * BK_FWD_WIN + wx;
            atomicMax(&win.px[k], key);
            if (offgrid) atomicMax(&win.tint[k], key);
        } else {
            const size_t o = (size_t)(ly - P.row0) * P.W + lx;
            atomicMax(&P.fwd_key_px[o], key);
            if (offgrid) atomicMax(&P.fwd_key_tint[o], key);
        }
    }
}
/* draw_quad (fisheye.c:2246-2338); int overflow on INT_MIN coordinates wraps as on x86-64.
 * A NaN projection becomes INT_MIN in uv_to_screen (cvttsd2si), and abs(INT_MIN - 0) is INT_MIN again: a quad with one bound at
 * INT_MIN and the other at exactly 0 PASSES the reference's 20-pixel size check and its loops then run over 2^31 rows or columns,
 * of which only those on the screen write anything.  The loops below visit the visible part of such a range only: bit-identical,
 * because a row's two crossings lie between corner x's - within the 20-pixel x extent, so no row of a tall quad can trip the
 * per-row abort - unless the x extent wraps as well (both bounds INT_MIN / 0 in x AND y: then the abort of an off-screen row is
 * not seen; the top-left pixel's neighbourhood under a lens that returns NaN for both coordinates). */
BK_DEV int bk_wrap_sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
/* two signed 16-bit lanes in an int: v_pk_min_i16 / v_pk_max_i16 on the device */
BK_DEV int bk_pack_i16(int x, int y) { return (int)(((unsigned int)x & 0xFFFFu) | ((unsigned int)y << 16)); }
#if defined(__clang__)
typedef short bk_s2 __attribute__((ext_vector_type(2)));
BK_DEV int bk_pk_min_i16(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(bk_s2, a), __builtin_bit_cast(bk_s2, b))); }
BK_DEV int bk_pk_max_i16(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(bk_s2, a), __builtin_bit_cast(bk_s2, b))); }
#else
BK_DEV int bk_pk_min_i16(int a, int b)
{
    const int ax = (short)(a & 0xFFFF), bx = (short)(b & 0xFFFF), ay = a >> 16, by = b >> 16;
    return bk_pack_i16(ax < bx ? ax : bx, ay < by ? ay : by);
}
BK_DEV int bk_pk_max_i16(int a, int b)
{
    const int ax = (short)(a & 0xFFFF), bx = (short)(b & 0xFFFF), ay = a >> 16, by = b >> 16;
    return bk_pack_i16(ax > bx ? ax : bx, ay > by ? ay : by);
}
#endif
BK_DEV int bk_imin(int a, int b) { return a < b ? a : b; }
BK_DEV int bk_imax(int a, int b) { return a > b ? a : b; }
BK_DEV void bk_draw_quad(const BkBuildParams &P, int x0, int y0, int x1, int y1, int x2, int y2, int x3, int y3,   /* tl, tr, br, bl: p[0..3] of :2251 */
                         unsigned int key, bool offgrid, int *wrote, const BkFwdWin &win)
{
    /* (the corners are values, and the edge loop below is written out: indexing an array of four corner pointers kept them in memory,
     * and every edge of every row waited for two loads) */
    int x = x0, y = y0;
    const int minx = bk_imin(bk_imin(x0, x1), bk_imin(x2, x3)), maxx = bk_imax(bk_imax(x0, x1), bk_imax(x2, x3));       /* :2257-2268 */
    const int miny = bk_imin(bk_imin(y0, y1), bk_imin(y2, y3)), maxy = bk_imax(bk_imax(y0, y1), bk_imax(y2, y3));
    const int maxdiff = 20;
    {
        int dx = bk_wrap_sub(minx, maxx), dy = bk_wrap_sub(miny, maxy);
        int adx = dx < 0 ? bk_wrap_sub(0, dx) : dx, ady = dy < 0 ? bk_wrap_sub(0, dy) : dy;      /* abs(): INT_MIN stays INT_MIN */
        if (adx > maxdiff || ady > maxdiff) return;                          /* :2272 */
    }
    /* A quad on one or two rows - every quad of a minifying lens: a 4K screen from six 2160-texel plates is 3.4 texels per pixel -
     * needs none of the arithmetic below.  Row miny: no edge has an end above it, the reference finds no crossing and fills
     * [minx, maxx] (:2304-2331 with tx[] as initialised).  Row maxy = miny + 1: an edge crosses it iff its ends lie on different rows,
     * and the interpolation (:2313) is (double)ix + 1.0 * dx or (double)ix + (+-0.0) * dx - exactly the x of the end that lies ON the
     * row; the first two such edges in the reference's order give the span.  The one-pixel, one-row and one-column cases (:2276-2301)
     * are the same two spans.  (Spans compared as wrapped differences: INT_MIN corners never pass.) */
    if ((unsigned int)bk_wrap_sub(maxx, minx) <= (unsigned int)maxdiff && (unsigned int)bk_wrap_sub(maxy, miny) <= 1u) {
        int t0 = minx, t1 = maxx, txi = 0;
#define BK_QUAD_EDGE2(ix, iy, jx, jy)                                       \
        if (txi < 2 && iy != jy) {                                          \
            const int t_ = iy == maxy ? ix : jx;                            \
            if (txi == 0) t0 = t_; else t1 = t_;                            \
            ++txi;                                                          \
        }
        BK_QUAD_EDGE2(x0, y0, x3, y3)
        BK_QUAD_EDGE2(x1, y1, x0, y0)
        BK_QUAD_EDGE2(x2, y2, x1, y1)
        BK_QUAD_EDGE2(x3, y3, x2, y2)
#undef BK_QUAD_EDGE2
        if (t0 > t1) { const int t = t0; t0 = t1; t1 = t; }
        const int nr = maxy - miny;
        for (int r = 0; r <= nr; ++r) bk_fwd_row(P, r ? t0 : minx, r ? t1 : maxx, r ? maxy : miny, key, offgrid, wrote, win);
        return;
    }
    /* the part of [min, max] that is on the screen (all of a normal, <= 21-long range that matters; one end of a 2^31-long one) */
    const int vx0 = minx < 0 ? 0 : minx, vx1 = maxx >= P.W ? P.W - 1 : maxx;
    const int vy0 = miny < 0 ? 0 : miny, vy1 = maxy >= P.H ? P.H - 1 : maxy;
    /* (a single pixel - :2276 - was a one-row quad above; what is left of :2283-2301 are lines of three pixels and more, and the 2^31-long ones) */
    if (miny == maxy) { bk_fwd_row(P, vx0, vx1, miny, key, offgrid, wrote, win); return; }
    if (minx == maxx) { for (int ty = vy0; ty <= vy1; ++ty) bk_fwd_row(P, x, x, ty, key, offgrid, wrote, win); return; }
    const bool tall = bk_wrap_sub(maxy, miny) < 0;                            /* 2^31 rows: visible ones only (see above) */
    const int y_first = tall ? vy0 : miny, nrows = bk_wrap_sub(tall ? vy1 : maxy, y_first);      /* <= 20, or <= H - 1; < 0: none */
    for (int ky = 0; ky <= nrows; ++ky) {
        y = (int)((unsigned)y_first + (unsigned)ky);
        int t0 = minx, t1 = maxx, txi = 0;
        /* edges in the reference's order - (p[0], p[3]), (p[1], p[0]), (p[2], p[1]), (p[3], p[2]) - until two of them cross the row */
#define BK_QUAD_EDGE(ix, iy, jx, jy)                                                                                              \
        if (txi < 2 && ((iy < y && y <= jy) || (jy < y && y <= iy))) {                                            /* :2310 */      \
            const double dx_ = (double)bk_wrap_sub(jx, ix);                                                                        \
            const int t_ = bk_trunc_to_int((double)ix + bk_edge_quot(bk_wrap_sub(y, iy), bk_wrap_sub(jy, iy), win.quot) * dx_);    /* :2313 */ \
            if (txi == 0) t0 = t_; else t1 = t_;                                                                                   \
            ++txi;                                                                                                                 \
        }
        BK_QUAD_EDGE(x0, y0, x3, y3)
        BK_QUAD_EDGE(x1, y1, x0, y0)
        BK_QUAD_EDGE(x2, y2, x1, y1)
        BK_QUAD_EDGE(x3, y3, x2, y2)
#undef BK_QUAD_EDGE
        if (t0 > t1) { const int t = t0; t0 = t1; t1 = t; }
        if (bk_wrap_sub(t1, t0) > maxdiff) return;                           /* :2327 aborts the quad */
        bk_fwd_row(P, t0, t1, y, key, offgrid, wrote, win);
    }
}

#ifndef BK_HAS_GLOBE_PLATE
/* Which 16 x 16 tiles of plate texels lie wholly inside their plate's own region, by a margin no rounding can bridge: one thread per
 * tile, grid (ceil(nt * nt / 256), plates).  A texel selects the plate whose forward vector has the largest float dot product with its
 * NORMALISED ray (fisheye.c:2023-2050; bk_ray_to_plate_index).  Before normalisation the ray is dist * forward + fu * right + fv * up,
 * affine in (fu, fv), so D_p - D_j - the un-normalised dot with the own plate's forward minus that with another's - is affine over the
 * tile and at least its minimum over the tile's four corner texels everywhere inside.  If that minimum is >= 1e-4 * Rmax * Fmax
 * (Rmax >= any ray's length on the plate, Fmax = the longest forward vector) the normalised dots differ by >= 1e-4 * Fmax, two hundred
 * times what the float operations of the reference's path can move them (a few 2^-24 * Fmax): every texel of the tile owns its ray,
 * strictly, and D_p >= 0 keeps the winning dot above the -2 the reference's search starts from.  Anything else - the tiles along a
 * plate's border, NaNs - is left to the exact test, texel by texel. */
extern "C" __global__ __launch_bounds__(256) void bk_forward_tiles(BkBuildParams P, unsigned char *tile_own)
{
    const int nt = (P.ps + BK_FWD_TILE - 1) / BK_FWD_TILE;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x), plate = (int)blockIdx.y;
    if (t >= nt * nt) return;
    const int ty = t / nt, tx = t - ty * nt;
    const BkPlateDev &p = P.plates[plate];
    double fmax = 0.0;
    for (int i = 0; i < P.numplates; ++i) {
        const float *f = P.plates[i].forward;
        const double l = __builtin_sqrt((double)f[0] * f[0] + (double)f[1] * f[1] + (double)f[2] * f[2]);
        fmax = l > fmax ? l : fmax;
    }
    const double lf = __builtin_sqrt((double)p.forward[0] * p.forward[0] + (double)p.forward[1] * p.forward[1] + (double)p.forward[2] * p.forward[2]);
    const double lr = __builtin_sqrt((double)p.right[0] * p.right[0] + (double)p.right[1] * p.right[1] + (double)p.right[2] * p.right[2]);
    const double lu = __builtin_sqrt((double)p.up[0] * p.up[0] + (double)p.up[1] * p.up[1] + (double)p.up[2] * p.up[2]);
    const double rmax = __builtin_fabs((double)p.dist) * lf + 0.5 * (lr + lu);
    const double margin = 1e-4 * rmax * fmax;
    bool all = margin > 0.0 && margin < BKM_INF;
    for (int c = 0; c < 4; ++c) {
        int x = tx * BK_FWD_TILE + ((c & 1) ? BK_FWD_TILE - 1 : 0), y = ty * BK_FWD_TILE + ((c & 2) ? BK_FWD_TILE - 1 : 0);
        x = x < P.ps ? x : P.ps - 1;
        y = y < P.ps ? y : P.ps - 1;
        const double fu = (double)P.fwd_uv[P.ps + 1 + x], fv = -(double)P.fwd_uv[P.ps + 1 + y];
        double r[3];
        for (int k = 0; k < 3; ++k) r[k] = (double)p.dist * p.forward[k] + fu * p.right[k] + fv * p.up[k];
        const double own = r[0] * p.forward[0] + r[1] * p.forward[1] + r[2] * p.forward[2];
        all = all && own >= 0.0;
        for (int j = 0; j < P.numplates; ++j) {
            if (j == plate) continue;
            const float *f = P.plates[j].forward;
            all = all && own - (r[0] * f[0] + r[1] * f[1] + r[2] * f[2]) >= margin;
        }
    }
    tile_own[((size_t)plate * nt + ty) * nt + tx] = all ? 1 : 0;
}
#endif

/* the quad loop of resume_lensmap_forward (fisheye.c:2189-2202): one thread per plate texel.
 * The reference writes plate-major, py descending, px ascending, later writers overwriting;
 * key = 1 + that sequence number, committed with atomicMax, reproduces the final state. */
/* (eight waves per SIMD - 64 VGPRs, the few beyond that spilled in the general scanline path: once the instruction count was down the
 *  pass was waiting on its own chain of loads and barriers, and the eighth wave is worth 17 % - profiles/r06_build_counters.txt, step l) */
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void bk_forward_quads(BkBuildParams P)      /* grid (ceil(ps/16), ceil(ps/16), plates): a tile of 16 x 16 texels */
{
    __shared__ int s_disp[6];
    __shared__ int s_box[4];                         /* the tile's bounding box on the screen: min x, min y, max x, max y */
    __shared__ unsigned int s_px[BK_FWD_WIN * BK_FWD_WIN], s_tint[BK_FWD_WIN * BK_FWD_WIN];
    const int tid = (int)threadIdx.x;
    if (tid < 6) s_disp[tid] = 0;
    if (tid < 2) s_box[tid] = 0x7FFFFFFF;
    else if (tid < 4) s_box[tid] = -1;
    __syncthreads();
    const int px = (int)blockIdx.x * BK_FWD_TILE + (tid & (BK_FWD_TILE - 1)), py = (int)blockIdx.y * BK_FWD_TILE + tid / BK_FWD_TILE, plate = (int)blockIdx.z;
    int err = 0;
    bool have = false;                               /* this texel owns its ray and all four of its corners project */
    int q[8] = {0, 0, 0, 0, 0, 0, 0, 0};             /* its corners on the screen: tl, tr, bl, br (x, y each) */
    int mx = 0x7FFFFFFF, my = 0x7FFFFFFF, Mx = -1, My = -1;       /* this lane's vote for the box (none: the neutral elements) */
    if (px < P.ps && py < P.ps) {
        const unsigned int id = ((unsigned int)plate * (unsigned int)P.ps + (unsigned int)py) * (unsigned int)P.ps + (unsigned int)px;
        /* the corners are asked for first - all of them, whether or not the texel turns out to own its ray: one round trip to memory,
         * spent under the arithmetic of the ownership test */
        const int n1 = P.ps + 1;
        const unsigned int c_tl = ((unsigned int)plate * (unsigned int)n1 + (unsigned int)py) * (unsigned int)n1 + (unsigned int)px, c_bl = c_tl + (unsigned int)n1;   /* (< 2^27 at 8K) */
        const unsigned int ok4 = (unsigned int)P.corner_ok[c_tl] & (unsigned int)P.corner_ok[c_tl + 1] & (unsigned int)P.corner_ok[c_bl] & (unsigned int)P.corner_ok[c_bl + 1];
        for (int c = 0; c < 4; ++c) { q[c] = P.corner_xy[2 * c_tl + c]; q[4 + c] = P.corner_xy[2 * c_bl + c]; }
        BkState S;
        bk_state_init(S, &P);
        /* (a tile bk_forward_tiles found inside its plate's region: uniform over the workgroup, the whole test is jumped over) */
        const bool tile_owns = P.tile_own && P.tile_own[((size_t)plate * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x];
        bool own = tile_owns || bk_texel_owns_at(P, S, plate, py, px);
#ifdef BK_HAS_GLOBE_PLATE
        if (P.ovr_count) {                          /* second pass: the host's answers for the texels the first pass flagged */
            unsigned int lo = 0, hi = P.ovr_count;
            while (lo < hi) {
                const unsigned int mid = (lo + hi) >> 1;
                if ((P.ovr_list[mid] >> 1) < id) lo = mid + 1; else hi = mid;
            }
            if (lo < P.ovr_count && (P.ovr_list[lo] >> 1) == id) own = P.ovr_list[lo] & 1u;
        } else if (S.flag) bk_push_flagged(P, id, own ? 1u : 0u, 0u, 0u);
#endif
        if (own && ok4) {
            have = true;
            /* the window covers the tile's quads from its top-left-most pixel on (corners that are nowhere near a screen - a NaN
             * projection is INT_MIN - do not vote) */
            const int a0 = bk_imin(bk_imin(q[0], q[2]), bk_imin(q[4], q[6])), b0 = bk_imax(bk_imax(q[0], q[2]), bk_imax(q[4], q[6]));
            const int a1 = bk_imin(bk_imin(q[1], q[3]), bk_imin(q[5], q[7])), b1 = bk_imax(bk_imax(q[1], q[3]), bk_imax(q[5], q[7]));
            if (a0 > -(1 << 24) && a1 > -(1 << 24) && b0 < (1 << 24) && b1 < (1 << 24)) { mx = a0 < 0 ? 0 : a0; my = a1 < 0 ? 0 : a1; Mx = b0; My = b1; }
        }
        err = S.err;
    }
    /* the votes are settled inside the wave first: 256 lanes of a tile on four LDS words cost more than everything they decide saves
     * (measured: four contended LDS atomics per thread made the pass 0.4 ms longer); eight waves' worth do not */
    /* ... as two pairs of 16-bit lanes (v_pk_min_i16 / v_pk_max_i16): the box only decides which writes meet in LDS, so clamping it to
     * 32767 pixels costs nothing but the window of a screen wider than that */
    {
        const bool voted = mx != 0x7FFFFFFF;
        int lo = bk_pack_i16(mx > 32767 ? 32767 : mx, my > 32767 ? 32767 : my);
        int hi = bk_pack_i16(Mx > 32767 ? 32767 : Mx < -1 ? -1 : Mx, My > 32767 ? 32767 : My < -1 ? -1 : My);
        for (int o = 32; o > 0; o >>= 1) {
            const int ql = __shfl_xor(lo, o), qh = __shfl_xor(hi, o);
            lo = bk_pk_min_i16(lo, ql);
            hi = bk_pk_max_i16(hi, qh);
        }
        const unsigned long long any = __ballot(voted);
        if ((tid & 63) == 0 && any) {
            atomicMin(&s_box[0], (int)(short)(lo & 0xFFFF)); atomicMin(&s_box[1], lo >> 16);
            atomicMax(&s_box[2], (int)(short)(hi & 0xFFFF)); atomicMax(&s_box[3], hi >> 16);
        }
    }
    __syncthreads();
    BkFwdWin win;
    win.x0 = s_box[0]; win.y0 = s_box[1];
    win.px = win.x0 != 0x7FFFFFFF ? s_px : nullptr;
    win.tint = s_tint;
    win.quot = P.fwd_quot;                          /* (read where it lies: only quads on three rows or more - magnifying lenses - interpolate) */
    win.w = win.px ? s_box[2] - win.x0 + 1 : 0;
    win.h = win.px ? s_box[3] - win.y0 + 1 : 0;
    win.w = win.w < 0 ? 0 : win.w > BK_FWD_WIN ? BK_FWD_WIN : win.w;
    win.h = win.h < 0 ? 0 : win.h > BK_FWD_WIN ? BK_FWD_WIN : win.h;
    /* only the part of the window the tile can touch is cleared and written out: a tile of 16 x 16 texels under a minifying lens
     * covers a dozen pixels each way, not 48 */
    const int col = tid & 63, row0w = tid >> 6;
    if (col < win.w)
        for (int wy = row0w; wy < win.h; wy += 4) { s_px[wy * BK_FWD_WIN + col] = 0u; s_tint[wy * BK_FWD_WIN + col] = 0u; }
    __syncthreads();
    if (have) {
        const unsigned int key = ((unsigned int)plate * (unsigned int)P.ps + (unsigned int)(P.ps - 1 - py)) * (unsigned int)P.ps + (unsigned int)px + 1u;
        int wrote = 0;
        bk_draw_quad(P, q[0], q[1], q[2], q[3], q[6], q[7], q[4], q[5], key, bk_offgrid(P, px, py), &wrote, win);
        if (wrote) s_disp[plate] = 1;
    }
    __syncthreads();
    if (col < win.w)                                 /* the window's pixels, once each (they passed bk_fwd_row's tests when they went in) */
        for (int wy = row0w; wy < win.h; wy += 4) {
            const unsigned int key = s_px[wy * BK_FWD_WIN + col];
            if (!key) continue;
            const size_t o = (size_t)(win.y0 + wy - P.row0) * P.W + (size_t)(win.x0 + col);
            atomicMax(&P.fwd_key_px[o], key);
            const unsigned int kt = s_tint[wy * BK_FWD_WIN + col];
            if (kt) atomicMax(&P.fwd_key_tint[o], kt);
        }
    bk_publish_flags(s_disp, P.display, err, P.err);
}

/* decode the winning keys into lens.pixels / lens.pixel_tints */
extern "C" __global__ __launch_bounds__(256) void bk_forward_resolve(BkBuildParams P)
{
    const size_t n = (size_t)P.rows * P.W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    /* the counters of the two passes before this one go to the host with the table: no copy command, no second stop (bk_lens.cpp) */
    if (P.counters_out && i < 18) P.counters_out[i] = P.display[i];
    if (i >= n) return;
    unsigned int off = 0xFFFFFFFFu;
    unsigned char tint = 255;
    const unsigned int k = P.fwd_key_px[i];
    if (k) {
        const unsigned int o = k - 1u, ps = (unsigned int)P.ps;
        const unsigned int px = o % ps, q = o / ps, plate = q / ps, py = ps - 1u - (q % ps);
        off = bk_padded_offset(P, (int)plate, (int)px, (int)py);
        const unsigned int kt = P.fwd_key_tint[i];
        if (kt) tint = (unsigned char)(((kt - 1u) / ps) / ps);
    }
    P.offsets[i] = off;
    P.tints[i] = tint;
}
#endif

/* The plate image f_saveglobe writes (WritePCXplate's pixel loop, fisheye.c:1438-1456): texel (j,i) of the
 * plate, or 0xFE where the ray through (u=j/ps, v=i/ps) belongs to another plate (ray_to_plate_index, which
 * may be the globe script's globe_plate) unless with_margins.  out = [ps][ps] row-major. */
extern "C" __global__ __launch_bounds__(256) void bk_save_plate(BkBuildParams P, int plate, int with_margins,
                                                                const unsigned char *globe_frame, unsigned char *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= P.ps || i >= P.ps) return;
    unsigned char col = globe_frame[bk_padded_offset(P, plate, j, i)];
    if (!with_margins) {
        const double v = ((double)i) / P.ps, u = ((double)j) / P.ps;                   /* :1439, 1441 */
        float ray[3];
        bk_plate_uv_to_ray(P, plate, u, v, ray);
        BkState S;
        bk_state_init(S, &P);
        const unsigned char texel = col;
        if (plate != bk_ray_to_plate_index(S, ray)) col = 0xFE;                        /* :1446 */
        if (S.flag) bk_push_flagged(P, (unsigned int)(i * P.ps + j), texel, 0u, 0u);   /* (a globe_plate script: the host decides) */
    }
    out[(size_t)i * P.ps + j] = col;
}

/* test / diagnosis hook: run one callback over an array of argument tuples and return the raw
 * double results (bk_debug_eval_device).  which: 0 lens_inverse, 1 lens_forward, 2 globe_plate.
 * nout[i] = number of results, -1 for a single nil, -100-err on a runtime error. */
extern "C" __global__ __launch_bounds__(256) void bk_eval_callback(BkBuildParams P, int which, const double *args,
                                                                   int nargs, int n, double *out, int *nout)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BkState S;
    bk_state_init(S, &P);
    bkv a[4], r[BK_MAXRET];
    for (int k = 0; k < 4; ++k) a[k] = k < nargs ? bk_num(args[(size_t)i * nargs + k]) : bk_nil();
    for (int k = 0; k < BK_MAXRET; ++k) r[k] = bk_nil();
    int m = 0;
#ifdef BK_HAS_INVERSE
    if (which == 0) m = LF_lens_inverse(S, a, nargs, r);
#endif
#ifdef BK_HAS_FORWARD
    if (which == 1) m = LF_lens_forward(S, a, nargs, r);
#endif
#ifdef BK_HAS_GLOBE_PLATE
    if (which == 2) m = LF_globe_plate(S, a, nargs, r);
#endif
    for (int k = 0; k < BK_MAXRET; ++k) out[(size_t)i * BK_MAXRET + k] = (k < m && r[k].t == BK_TNUM) ? r[k].n : __builtin_nan("");
    nout[i] = S.err ? -100 - S.err : (m == 1 && r[0].t == BK_TNIL) ? -1 : m;
}
#endif  /* !BK_HOST_MODULE */
