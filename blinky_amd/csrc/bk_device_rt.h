/* bk_device_rt.h -- device-side runtime of the generated lens code (embedded into the hiprtc
 * translation unit after bkm.h and bk_build_params.h).  A Lua value on the device is a tagged
 * double; after inlining LLVM folds almost every tag test away.  Mutable script globals are
 * per-thread fields of BkState (declared by the emitter through BK_MUTABLE_GLOBALS). */
typedef struct { double n; int t; } bkv;
#define BK_TNIL 0
#define BK_TFALSE 1
#define BK_TTRUE 2
#define BK_TNUM 3
#define BK_MAXRET 8
#define BK_LOOP_BUDGET (1 << 22)
#define BK_DEV static __device__ __forceinline__

struct BkState {
    const BkBuildParams *P;
    int err;
    int steps;
    BK_MUTABLE_GLOBALS
};

BK_DEV bkv bk_num(double d) { bkv v; v.n = d; v.t = BK_TNUM; return v; }
BK_DEV bkv bk_nil() { bkv v; v.n = 0.0; v.t = BK_TNIL; return v; }
BK_DEV bkv bk_bool(bool b) { bkv v; v.n = 0.0; v.t = b ? BK_TTRUE : BK_TFALSE; return v; }
BK_DEV bool bk_truthy(bkv v) { return v.t >= BK_TTRUE; }
BK_DEV bool bk_isnum(bkv v) { return v.t == BK_TNUM; }
BK_DEV double bk_tonum(BkState &S, bkv v) { if (v.t != BK_TNUM) S.err |= BK_ERR_ARITH; return v.n; }
BK_DEV bool bk_tick(BkState &S) { if (++S.steps > BK_LOOP_BUDGET) { S.err |= BK_ERR_LOOP; return false; } return true; }

BK_DEV bkv bk_add(BkState &S, bkv a, bkv b) { return bk_num(bk_tonum(S, a) + bk_tonum(S, b)); }
BK_DEV bkv bk_sub(BkState &S, bkv a, bkv b) { return bk_num(bk_tonum(S, a) - bk_tonum(S, b)); }
BK_DEV bkv bk_mul(BkState &S, bkv a, bkv b) { return bk_num(bk_tonum(S, a) * bk_tonum(S, b)); }
BK_DEV bkv bk_div(BkState &S, bkv a, bkv b) { return bk_num(bk_tonum(S, a) / bk_tonum(S, b)); }
BK_DEV bkv bk_mod(BkState &S, bkv a, bkv b)       /* luai_nummod: a - floor(a/b)*b */
{
    double x = bk_tonum(S, a), y = bk_tonum(S, b);
    return bk_num(x - bkm_floor(x / y) * y);
}
BK_DEV bkv bk_pow(BkState &S, bkv a, bkv b) { return bk_num(bkm_pow(bk_tonum(S, a), bk_tonum(S, b))); }
BK_DEV bkv bk_unm(BkState &S, bkv a) { return bk_num(-bk_tonum(S, a)); }
BK_DEV bkv bk_not(bkv a) { return bk_bool(!bk_truthy(a)); }
BK_DEV bkv bk_eq(bkv a, bkv b) { return bk_bool(a.t == b.t && (a.t != BK_TNUM || a.n == b.n)); }
BK_DEV bkv bk_ne(bkv a, bkv b) { return bk_bool(!(a.t == b.t && (a.t != BK_TNUM || a.n == b.n))); }
BK_DEV bkv bk_lt(BkState &S, bkv a, bkv b)
{
    if (a.t != BK_TNUM || b.t != BK_TNUM) S.err |= BK_ERR_COMPARE;
    return bk_bool(a.n < b.n);
}
BK_DEV bkv bk_le(BkState &S, bkv a, bkv b)
{
    if (a.t != BK_TNUM || b.t != BK_TNUM) S.err |= BK_ERR_COMPARE;
    return bk_bool(a.n <= b.n);
}
/* 1-based arrays (local array tables / snapshots of global array tables) */
BK_DEV bkv bk_aget(const bkv *arr, int n, bkv idx)
{
    if (idx.t == BK_TNUM && idx.n >= 1.0 && idx.n <= (double)n && idx.n == bkm_trunc(idx.n)) return arr[(int)idx.n];
    return bk_nil();
}
BK_DEV void bk_aset(BkState &S, bkv *arr, int n, bkv idx, bkv v)
{
    if (idx.t == BK_TNUM && idx.n >= 1.0 && idx.n <= (double)n && idx.n == bkm_trunc(idx.n)) arr[(int)idx.n] = v;
    else S.err |= BK_ERR_INDEX;
}

/* ---- mathlib.c / fisheye.c helpers, float arithmetic exactly as the reference (no FMA) ---- */
BK_DEV float bk_dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }   /* mathlib.h:70 */
BK_DEV void bk_vector_ma(const float *a, float scale, const float *b, float *c)                      /* mathlib.c:350 */
{
    c[0] = a[0] + scale * b[0];
    c[1] = a[1] + scale * b[1];
    c[2] = a[2] + scale * b[2];
}
BK_DEV void bk_vector_normalize(float *v)                                                            /* mathlib.c:413 */
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
BK_DEV void bk_latlon_to_ray(double lat, double lon, float *ray)                                     /* fisheye.c:1184 */
{
    double clat = bkm_cos(lat);
    ray[0] = (float)(bkm_sin(lon) * clat);
    ray[1] = (float)bkm_sin(lat);
    ray[2] = (float)(bkm_cos(lon) * clat);
}
BK_DEV void bk_ray_to_latlon(const float *ray, double *lat, double *lon)                             /* fisheye.c:1192 */
{
    *lon = bkm_atan2((double)ray[0], (double)ray[2]);
    *lat = bkm_atan2((double)ray[1], bkm_sqrt((double)(ray[0] * ray[0] + ray[2] * ray[2])));
}
BK_DEV void bk_plate_uv_to_ray(const BkBuildParams &P, int plate, double u, double v, float *ray)    /* fisheye.c:1198 */
{
    const BkPlateDev &p = P.plates[plate];
    u -= 0.5;
    v -= 0.5;
    v = -v;
    ray[0] = ray[1] = ray[2] = 0;
    bk_vector_ma(ray, p.dist, p.forward, ray);
    bk_vector_ma(ray, (float)u, p.right, ray);
    bk_vector_ma(ray, (float)v, p.up, ray);
    bk_vector_normalize(ray);
}
/* (int)double the way x86-64 cvttsd2si does: NaN / out of range -> INT_MIN */
BK_DEV int bk_trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

/* the C functions the reference registers for scripts (fisheye.c:1494-1537) */
BK_DEV int bk_host_latlon_to_ray(BkState &S, bkv lat, bkv lon, bkv *r)
{
    float ray[3];
    bk_latlon_to_ray(bk_tonum(S, lat), bk_tonum(S, lon), ray);
    r[0] = bk_num((double)ray[0]); r[1] = bk_num((double)ray[1]); r[2] = bk_num((double)ray[2]);
    return 3;
}
BK_DEV int bk_host_ray_to_latlon(BkState &S, bkv x, bkv y, bkv z, bkv *r)
{
    float ray[3] = {(float)bk_tonum(S, x), (float)bk_tonum(S, y), (float)bk_tonum(S, z)};
    double lat, lon;
    bk_ray_to_latlon(ray, &lat, &lon);
    r[0] = bk_num(lat); r[1] = bk_num(lon);
    return 2;
}
BK_DEV int bk_host_plate_to_ray(BkState &S, bkv plate, bkv u, bkv v, bkv *r)
{
    int pi = (int)bk_tonum(S, plate);            /* int plate_index = luaL_checknumber(...)  :1523 */
    float ray[3];
    if (pi < 0 || pi >= S.P->numplates) { r[0] = bk_nil(); return 1; }
    bk_plate_uv_to_ray(*S.P, pi, bk_tonum(S, u), bk_tonum(S, v), ray);
    r[0] = bk_num((double)ray[0]); r[1] = bk_num((double)ray[1]); r[2] = bk_num((double)ray[2]);
    return 3;
}
