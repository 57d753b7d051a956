/* bk_device_rt.h -- device-side runtime of the generated lens code (embedded into the hiprtc
 * translation unit after bkm.h and bk_build_params.h).  A Lua value on the device is a tagged
 * double; after inlining LLVM folds almost every tag test away.  Mutable script globals are
 * per-thread fields of BkState (declared by the emitter through BK_MUTABLE_GLOBALS).
 *
 * Exactness against the reference's platform libm.  Every IEEE operation here is the one the
 * reference performs, so the only place where this code and a Lua VM on glibc can disagree is the
 * last bits of a transcendental (bkm.h: <= 0.52 ulp; glibc: <= 2 ulp documented).  Each value
 * therefore carries `e`, a running bound on |n - n_ref| where n_ref is what the same script
 * computes on ANY libm whose results lie within BK_LIBM_REL of bkm.h's: 0 for everything derived
 * from the arguments by IEEE operations alone, |f(x)| * BK_LIBM_REL after a libm call, propagated
 * through arithmetic by the usual first-order bounds plus one rounding.  Wherever a value is used
 * DISCRETELY - narrowed to float (fisheye.c:1187-1189, 1559-1561), compared, floored, truncated,
 * used as an index - and its interval [n-e, n+e] straddles the decision, the pixel is flagged
 * (S.flag).  Unflagged pixels are provably what the reference computes; the flagged ones (a handful
 * per 4K map) are re-evaluated by the host interpreter on the platform libm (bk_lens.cpp,
 * fixup_flagged) and patched.  `n` itself is never touched by the bookkeeping. */
typedef struct { double n; double e; int t; } bkv;
#define BK_TNIL 0
#define BK_TFALSE 1
#define BK_TTRUE 2
#define BK_TNUM 3
#define BK_TSTR 4                    /* a string CONSTANT of the script: n = its number in the emitter's intern table (equality only) */
#define BK_MAXRET 8
#define BK_LOOP_BUDGET (1 << 22)
#define BK_DEV static __device__ __forceinline__
#ifndef BK_LIBM_REL                  /* (tests/hostemu widens it to check the propagation rules) */
#define BK_LIBM_REL 0x1p-50          /* assumed bound on |libm_ref(x) - bkm(x)| / |bkm(x)| (4 x 2^-52) */
#endif
#define BK_ROUND_REL 0x1p-51         /* one rounding on each side of an operation with inexact inputs */

struct BkState {
    const BkBuildParams *P;
    int err;
    int steps;
    int flag;                        /* a discrete decision of this pixel depends on libm's last bits */
    BK_MUTABLE_GLOBALS
};

BK_DEV bkv bk_num(double d) { bkv v; v.n = d; v.e = 0.0; v.t = BK_TNUM; return v; }
BK_DEV bkv bk_nume(double d, double e) { bkv v; v.n = d; v.e = e; v.t = BK_TNUM; return v; }
BK_DEV bkv bk_nil() { bkv v; v.n = 0.0; v.e = 0.0; v.t = BK_TNIL; return v; }
BK_DEV bkv bk_bool(bool b) { bkv v; v.n = 0.0; v.e = 0.0; v.t = b ? BK_TTRUE : BK_TFALSE; return v; }
BK_DEV bkv bk_str(int id) { bkv v; v.n = (double)id; v.e = 0.0; v.t = BK_TSTR; return v; }
/* type(v): interned ids 1..4 are reserved for "nil", "boolean", "number", "string" (bk_emit.cpp) */
BK_DEV bkv bk_typeof(bkv v) { return bk_str(v.t == BK_TNIL ? 1 : v.t == BK_TNUM ? 3 : v.t == BK_TSTR ? 4 : 2); }
BK_DEV bool bk_truthy(bkv v) { return v.t >= BK_TTRUE; }
BK_DEV bool bk_isnum(bkv v) { return v.t == BK_TNUM; }
BK_DEV double bk_tonum(BkState &S, bkv v) { if (v.t != BK_TNUM) S.err |= BK_ERR_ARITH; return v.n; }
BK_DEV bool bk_tick(BkState &S) { if (++S.steps > BK_LOOP_BUDGET) { S.err |= BK_ERR_LOOP; return false; } return true; }

/* ---- error bookkeeping helpers ------------------------------------------------------------------ */
BK_DEV double bk_abs(double x) { return __builtin_fabs(x); }
/* A NaN or infinity that arises from exact arguments arises on every libm alike (domain errors, overflow, x/0) and then
 * propagates by IEEE rules: it carries no bound.  One that arises from INEXACT arguments cannot be bounded: flagged. */
BK_DEV bool bk_finite(double z) { return bk_abs(z) < BKM_INF; }
/* bound after an IEEE operation whose inputs are inexact: propagated part + the two roundings (one fma; its result is finite
 * exactly when z and the propagated part are - overflow aside, which is flagged like them) */
BK_DEV double bk_eop(BkState &S, double z, double eprop)
{
    if (eprop == 0.0) return 0.0;
    const double r = __builtin_fma(bk_abs(z), BK_ROUND_REL, eprop);
    if (!(r < BKM_INF)) { S.flag = 1; return 0.0; }
    return r;
}
/* bound after a libm call: propagated part + the libm discrepancy itself (also for exact inputs) */
BK_DEV double bk_elibm(BkState &S, double z, double eprop)
{
    const double r = __builtin_fma(bk_abs(z), BK_LIBM_REL, eprop);
    if (!(r < BKM_INF)) { if (eprop != 0.0) S.flag = 1; return 0.0; }
    return r;
}
/* A step of a self-correcting iteration (bk_emit.cpp, contraction_pattern): the variable carried from step to step entered this
 * step with bound e0 and was taken as exact inside it; `d` is the derivative of the value this bound belongs to with respect to
 * the carried variable, `el` its bound within the step.  First order in e0 with a factor of two of slack, plus e0 * 2^20 * (the
 * libm bound) for what first order leaves out where d vanishes: the step's own libm errors are priced at THIS side's value of the
 * carried variable, the other side's is e0 away (a Newton step that lands on a root at 0 exactly computes 0 with el = 0, where
 * a libm one ulp off leaves 4 ulp * e0), and d itself moves by (second derivative) * e0.  Both are e0 times a condition number
 * times something small; 2^20 covers condition numbers of a million.  Refused (flagged) where the incoming bound is too wide for
 * any of this to mean anything. */
BK_DEV double bk_contract(BkState &S, double d, double e0, double el)
{
    if (e0 == 0.0) return el;
    const double r = __builtin_fma(__builtin_fma(2.0, bk_abs(d), 0x1p20 * BK_LIBM_REL), e0, el);
    if (!(e0 < 0x1p20 * BK_LIBM_REL) || !(r < BKM_INF)) { S.flag = 1; return el; }     /* (2^-30 with the libm bound of 2^-50) */
    return r;
}
/* the decision "which integer is floor/trunc/rint of x" is stable over [x-e, x+e] */
BK_DEV void bk_need_same_floor(BkState &S, double x, double e)
{
    if (e != 0.0 && !(bkm_floor(x - e) == bkm_floor(x + e))) S.flag = 1;
}
BK_DEV void bk_need_same_trunc(BkState &S, double x, double e)
{
    if (e != 0.0 && !(bkm_trunc(x - e) == bkm_trunc(x + e))) S.flag = 1;
}
BK_DEV void bk_need_exact(BkState &S, bkv v) { if (v.e != 0.0) S.flag = 1; }
/* (float)v as the reference narrows a Lua number into a vec_t; flagged when the interval crosses a rounding tie */
BK_DEV float bk_narrow(BkState &S, double v, double e)
{
    const float f = (float)v;
    if (e != 0.0 && !((float)(v - e) == f && (float)(v + e) == f) && v == v) S.flag = 1;
    return f;
}

BK_DEV bkv bk_add(BkState &S, bkv a, bkv b)
{
    const double z = bk_tonum(S, a) + bk_tonum(S, b);
    return bk_nume(z, bk_eop(S, z, a.e + b.e));
}
BK_DEV bkv bk_sub(BkState &S, bkv a, bkv b)
{
    const double z = bk_tonum(S, a) - bk_tonum(S, b);
    return bk_nume(z, bk_eop(S, z, a.e + b.e));
}
BK_DEV bkv bk_mul(BkState &S, bkv a, bkv b)
{
    const double x = bk_tonum(S, a), y = bk_tonum(S, b), z = x * y;
    if (a.e == 0.0 && b.e == 0.0) return bk_num(z);
    return bk_nume(z, bk_eop(S, z, bk_abs(x) * b.e + bk_abs(y) * a.e + a.e * b.e));
}
BK_DEV double bk_ediv(BkState &S, double x, double ex, double y, double ey, double z)
{
    if (ex == 0.0 && ey == 0.0) return 0.0;
    if (!(bk_abs(y) > 2.0 * ey)) { S.flag = 1; return 0.0; }        /* the divisor's sign / magnitude is not determined */
    return bk_eop(S, z, (ex + bk_abs(z) * ey) / (bk_abs(y) - ey));
}
BK_DEV bkv bk_div(BkState &S, bkv a, bkv b)
{
    const double x = bk_tonum(S, a), y = bk_tonum(S, b), z = x / y;
    return bk_nume(z, bk_ediv(S, x, a.e, y, b.e, z));
}
BK_DEV bkv bk_mod(BkState &S, bkv a, bkv b)       /* luai_nummod: a - floor(a/b)*b */
{
    const double x = bk_tonum(S, a), y = bk_tonum(S, b);
    const double q = x / y, fl = bkm_floor(q), z = x - fl * y;
    if (a.e == 0.0 && b.e == 0.0) return bk_num(z);
    bk_need_same_floor(S, q, bk_ediv(S, x, a.e, y, b.e, q));
    return bk_nume(z, bk_eop(S, z, a.e + bk_abs(fl) * b.e) + bk_abs(fl * y) * BK_ROUND_REL);
}
/* z = x ^ y through bkm_pow (glibc's pow on the reference side) */
BK_DEV bkv bk_powv(BkState &S, bkv a, bkv b)
{
    const double x = bk_tonum(S, a), y = bk_tonum(S, b), z = bkm_pow(x, y);
    double ep = 0.0;
    if (b.e == 0.0 && a.e != 0.0 && y == bkm_trunc(y) && y >= 1.0 && y <= 64.0) {
        /* an integer power is smooth through 0 and for negative bases: |d x^n| <= n (|x| + e)^(n-1) dx */
        double m = 1.0;
        for (int k = 1; k < (int)y; ++k) m *= bk_abs(x) + a.e;
        ep = 2.0 * y * m * a.e;
    } else if (a.e != 0.0 || b.e != 0.0) {
        /* d(x^y) = x^y (y dx/x + ln x dy); only for a base safely away from 0 and a finite result */
        if (!(x - 2.0 * a.e > 0.0) || !(bk_abs(z) < BKM_INF)) { S.flag = 1; return bk_num(z); }
        ep = 2.0 * bk_abs(z) * (bk_abs(y) * a.e / (x - a.e) + (bk_abs(bkm_log(x)) + 1.0) * b.e);
    }
    return bk_nume(z, bk_elibm(S, z, ep));
}
BK_DEV bkv bk_pow(BkState &S, bkv a, bkv b) { return bk_powv(S, a, b); }
BK_DEV bkv bk_unm(BkState &S, bkv a) { return bk_nume(-bk_tonum(S, a), a.e); }
BK_DEV bkv bk_not(bkv a) { return bk_bool(!bk_truthy(a)); }
/* a comparison is decided by libm's last bits when the two intervals overlap */
BK_DEV void bk_need_apart(BkState &S, bkv a, bkv b)
{
    const double e = a.e + b.e;
    if (e != 0.0 && !(bk_abs(a.n - b.n) > e)) S.flag = 1;
}
BK_DEV bkv bk_eq(BkState &S, bkv a, bkv b)
{
    if (a.t == BK_TNUM && b.t == BK_TNUM) bk_need_apart(S, a, b);
    return bk_bool(a.t == b.t && (a.t < BK_TNUM || a.n == b.n));
}
BK_DEV bkv bk_ne(BkState &S, bkv a, bkv b)
{
    if (a.t == BK_TNUM && b.t == BK_TNUM) bk_need_apart(S, a, b);
    return bk_bool(!(a.t == b.t && (a.t < BK_TNUM || a.n == b.n)));
}
BK_DEV bkv bk_lt(BkState &S, bkv a, bkv b)
{
    if (a.t != BK_TNUM || b.t != BK_TNUM) S.err |= BK_ERR_COMPARE;
    bk_need_apart(S, a, b);
    return bk_bool(a.n < b.n);
}
BK_DEV bkv bk_le(BkState &S, bkv a, bkv b)
{
    if (a.t != BK_TNUM || b.t != BK_TNUM) S.err |= BK_ERR_COMPARE;
    bk_need_apart(S, a, b);
    return bk_bool(a.n <= b.n);
}
/* 1-based arrays (local array tables / snapshots of global array tables) */
BK_DEV bkv bk_aget(BkState &S, const bkv *arr, int n, bkv idx)
{
    bk_need_exact(S, idx);
    if (idx.t == BK_TNUM && idx.n >= 1.0 && idx.n <= (double)n && idx.n == bkm_trunc(idx.n)) return arr[(int)idx.n];
    return bk_nil();
}
BK_DEV void bk_aset(BkState &S, bkv *arr, int n, bkv idx, bkv v)
{
    bk_need_exact(S, idx);
    if (idx.t == BK_TNUM && idx.n >= 1.0 && idx.n <= (double)n && idx.n == bkm_trunc(idx.n)) arr[(int)idx.n] = v;
    else S.err |= BK_ERR_INDEX;
}

/* ---- the math library (lmathlib.c on the reference side), value + bound --------------------------- */
BK_DEV bkv bk_f_sin(BkState &S, bkv a) { const double z = bkm_sin(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, a.e)); }
BK_DEV bkv bk_f_cos(BkState &S, bkv a) { const double z = bkm_cos(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, a.e)); }
/* math.sin(a) and math.cos(a) of one operand in one statement (bk_emit.cpp): one argument reduction for the two */
BK_DEV void bk_f_sincos(BkState &S, bkv a, bkv *s, bkv *c)
{
    double zs, zc;
    bkm_sincos(bk_tonum(S, a), &zs, &zc);
    *s = bk_nume(zs, bk_elibm(S, zs, a.e));
    *c = bk_nume(zc, bk_elibm(S, zc, a.e));
}
BK_DEV bkv bk_f_atan(BkState &S, bkv a) { const double z = bkm_atan(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, a.e)); }
BK_DEV bkv bk_f_tanh(BkState &S, bkv a) { const double z = bkm_tanh(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, a.e)); }
BK_DEV bkv bk_f_tan(BkState &S, bkv a)
{
    const double z = bkm_tan(bk_tonum(S, a));
    double ep = 0.0;
    if (a.e != 0.0) {
        ep = 2.0 * a.e * (1.0 + z * z);                                     /* sec^2, with slack for its change over the interval */
        if (!(ep < 0x1p-10 * (1.0 + bk_abs(z)))) { S.flag = 1; ep = 0.0; }
    }
    return bk_nume(z, bk_elibm(S, z, ep));
}
BK_DEV double bk_e_asin(BkState &S, double x, double e)                   /* Lipschitz bound of asin / acos over [x-e, x+e] */
{
    if (e == 0.0) return 0.0;
    const double m = bk_abs(x) + e;
    if (!(m < 1.0)) { S.flag = 1; return 0.0; }                            /* the interval reaches the domain's edge */
    return e / bkm_sqrt((1.0 - m) * (1.0 + m));
}
BK_DEV bkv bk_f_asin(BkState &S, bkv a)
{
    const double x = bk_tonum(S, a), z = bkm_asin(x);
    return bk_nume(z, bk_elibm(S, z, bk_e_asin(S, x, a.e)));
}
BK_DEV bkv bk_f_acos(BkState &S, bkv a)
{
    const double x = bk_tonum(S, a), z = bkm_acos(x);
    return bk_nume(z, bk_elibm(S, z, bk_e_asin(S, x, a.e)));
}
BK_DEV double bk_e_grow(BkState &S, double z, double e)                    /* sinh / cosh / exp: |f'| <= 1 + |f| */
{
    if (e == 0.0) return 0.0;
    if (!(e < 0x1p-10)) { S.flag = 1; return 0.0; }
    return 2.0 * e * (1.0 + bk_abs(z));
}
BK_DEV bkv bk_f_sinh(BkState &S, bkv a) { const double z = bkm_sinh(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, bk_e_grow(S, z, a.e))); }
BK_DEV bkv bk_f_cosh(BkState &S, bkv a) { const double z = bkm_cosh(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, bk_e_grow(S, z, a.e))); }
BK_DEV bkv bk_f_exp(BkState &S, bkv a) { const double z = bkm_exp(bk_tonum(S, a)); return bk_nume(z, bk_elibm(S, z, bk_e_grow(S, z, a.e))); }
BK_DEV double bk_e_log(BkState &S, double x, double e)
{
    if (e == 0.0) return 0.0;
    if (!(x - 2.0 * e > 0.0)) { S.flag = 1; return 0.0; }
    return e / (x - e);
}
BK_DEV bkv bk_f_log(BkState &S, bkv a) { const double x = bk_tonum(S, a), z = bkm_log(x); return bk_nume(z, bk_elibm(S, z, bk_e_log(S, x, a.e))); }
BK_DEV bkv bk_f_log10(BkState &S, bkv a) { const double x = bk_tonum(S, a), z = bkm_log10(x); return bk_nume(z, bk_elibm(S, z, bk_e_log(S, x, a.e))); }
BK_DEV bkv bk_f_sqrt(BkState &S, bkv a)                                    /* IEEE: exact on an exact argument */
{
    const double x = bk_tonum(S, a), z = bkm_sqrt(x);
    if (a.e == 0.0) return bk_num(z);
    if (!(x - 2.0 * a.e > 0.0)) { S.flag = 1; return bk_num(z); }
    return bk_nume(z, bk_eop(S, z, a.e / (2.0 * bkm_sqrt(x - a.e))));
}
BK_DEV bkv bk_f_abs(BkState &S, bkv a) { return bk_nume(bkm_fabs(bk_tonum(S, a)), a.e); }
BK_DEV bkv bk_f_floor(BkState &S, bkv a) { const double x = bk_tonum(S, a); bk_need_same_floor(S, x, a.e); return bk_num(bkm_floor(x)); }
BK_DEV bkv bk_f_ceil(BkState &S, bkv a) { const double x = bk_tonum(S, a); bk_need_same_floor(S, -x, a.e); return bk_num(bkm_ceil(x)); }
BK_DEV double bk_e_atan2(BkState &S, double y, double ey, double x, double ex)
{
    if (ex == 0.0 && ey == 0.0) return 0.0;
    const double d = x * x + y * y, w = bk_abs(x) * ey + bk_abs(y) * ex, s = ex + ey;
    /* the point must stay clear of the origin and of the branch cut along the negative x axis */
    if (!(d > 16.0 * s * s) || (x < 0.0 && !(bk_abs(y) > ey))) { S.flag = 1; return 0.0; }
    return 2.0 * w / d;
}
BK_DEV bkv bk_f_atan2(BkState &S, bkv a, bkv b)
{
    const double y = bk_tonum(S, a), x = bk_tonum(S, b), z = bkm_atan2(y, x);
    /* on an axis (exact arguments) every libm returns the correctly rounded 0, +-pi/2 or +-pi (C99 F.9.1.4) */
    if (a.e == 0.0 && b.e == 0.0 && (x == 0.0 || y == 0.0)) return bk_num(z);
    return bk_nume(z, bk_elibm(S, z, bk_e_atan2(S, y, a.e, x, b.e)));
}
BK_DEV bkv bk_f_fmod(BkState &S, bkv a, bkv b)                             /* C fmod: exact, x - trunc(x/y)*y */
{
    const double x = bk_tonum(S, a), y = bk_tonum(S, b), z = bkm_fmod(x, y);
    if (a.e == 0.0 && b.e == 0.0) return bk_num(z);
    const double q = x / y;
    bk_need_same_trunc(S, q, bk_ediv(S, x, a.e, y, b.e, q));
    return bk_nume(z, bk_eop(S, z, a.e + bk_abs(bkm_trunc(q)) * b.e));
}
BK_DEV bkv bk_f_scale(BkState &S, bkv a, double c, bool divide)            /* math.deg / math.rad */
{
    const double x = bk_tonum(S, a), z = divide ? x / c : x * c;
    return bk_nume(z, bk_eop(S, z, divide ? a.e / c : a.e * c));
}
BK_DEV bkv bk_f_logb(BkState &S, bkv a, bkv b)                             /* math.log(x [, base]), lmathlib.c */
{
    if (b.t == BK_TNIL) return bk_f_log(S, a);
    bk_need_exact(S, b);                                                   /* which formula is taken depends on the base */
    if (bk_tonum(S, b) == 10.0) return bk_f_log10(S, a);
    return bk_div(S, bk_f_log(S, a), bk_f_log(S, b));
}
BK_DEV bkv bk_f_pick(BkState &S, bkv m, bkv d, bool want_max)              /* one step of math.max / math.min (1-Lipschitz) */
{
    const double dn = bk_tonum(S, d);
    const bool take = want_max ? dn > m.n : dn < m.n;
    return bk_nume(take ? dn : m.n, m.e > d.e ? m.e : d.e);
}
BK_DEV void bk_f_modf(BkState &S, bkv a, bkv *r)
{
    const double x = bk_tonum(S, a), ip = bkm_trunc(x);
    bk_need_same_trunc(S, x, a.e);
    r[0] = bk_num(ip);
    r[1] = bk_nume(bkm_isinf(x) ? bkm_copysign(0.0, x) : x - ip, a.e);
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
/* latlon_to_ray, fisheye.c:1184: double products narrowed into a vec3_t; `elat`/`elon` bound the arguments */
BK_DEV void bk_latlon_to_ray(BkState &S, double lat, double elat, double lon, double elon, float *ray)
{
    double slat, clat, slon, clon;                     /* (one argument reduction per angle: bkm_sincos == bkm_sin, bkm_cos bit for bit) */
    bkm_sincos(lat, &slat, &clat);
    bkm_sincos(lon, &slon, &clon);
    const double ec = bk_elibm(S, clat, elat), es = bk_elibm(S, slon, elon), ek = bk_elibm(S, clon, elon);
    const double p0 = slon * clat, p2 = clon * clat;
    ray[0] = bk_narrow(S, p0, bk_eop(S, p0, bk_abs(slon) * ec + bk_abs(clat) * es + ec * es));
    ray[1] = bk_narrow(S, slat, bk_elibm(S, slat, elat));
    ray[2] = bk_narrow(S, p2, bk_eop(S, p2, bk_abs(clon) * ec + bk_abs(clat) * ek + ec * ek));
}
/* plate_uv_to_ray from the two floats VectorMA scales right and up by: fu = (float)(u - 0.5), fv = (float)(-(v - 0.5)) */
BK_DEV void bk_plate_fuv_to_ray(const BkBuildParams &P, int plate, float fu, float fv, float *ray)
{
    const BkPlateDev &p = P.plates[plate];
    ray[0] = ray[1] = ray[2] = 0;
    bk_vector_ma(ray, p.dist, p.forward, ray);
    bk_vector_ma(ray, fu, p.right, ray);
    bk_vector_ma(ray, fv, p.up, ray);
    bk_vector_normalize(ray);
}
BK_DEV void bk_plate_uv_to_ray(const BkBuildParams &P, int plate, double u, double v, float *ray)    /* fisheye.c:1198 */
{
    u -= 0.5;
    v -= 0.5;
    v = -v;
    bk_plate_fuv_to_ray(P, plate, (float)u, (float)v, ray);
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
    bk_latlon_to_ray(S, bk_tonum(S, lat), lat.e, bk_tonum(S, lon), lon.e, ray);
    r[0] = bk_num((double)ray[0]); r[1] = bk_num((double)ray[1]); r[2] = bk_num((double)ray[2]);
    return 3;
}
BK_DEV int bk_host_ray_to_latlon(BkState &S, bkv x, bkv y, bkv z, bkv *r)      /* fisheye.c:1192, 1506-1519 */
{
    const float ray[3] = {bk_narrow(S, bk_tonum(S, x), x.e), bk_narrow(S, bk_tonum(S, y), y.e), bk_narrow(S, bk_tonum(S, z), z.e)};
    const double lon = bkm_atan2((double)ray[0], (double)ray[2]);
    const double lat = bkm_atan2((double)ray[1], bkm_sqrt((double)(ray[0] * ray[0] + ray[2] * ray[2])));
    r[0] = bk_nume(lat, bk_elibm(S, lat, 0.0)); r[1] = bk_nume(lon, bk_elibm(S, lon, 0.0));
    return 2;
}
BK_DEV int bk_host_plate_to_ray(BkState &S, bkv plate, bkv u, bkv v, bkv *r)
{
    bk_need_same_trunc(S, bk_tonum(S, plate), plate.e);
    int pi = (int)bk_tonum(S, plate);            /* int plate_index = luaL_checknumber(...)  :1523 */
    float ray[3];
    if (pi < 0 || pi >= S.P->numplates) { r[0] = bk_nil(); return 1; }
    /* u, v are narrowed by VectorMA's float scale after the double subtraction of 0.5 (:1209-1211) */
    (void)bk_narrow(S, bk_tonum(S, u) - 0.5, u.e);
    (void)bk_narrow(S, -(bk_tonum(S, v) - 0.5), v.e);
    bk_plate_uv_to_ray(*S.P, pi, u.n, v.n, ray);
    r[0] = bk_num((double)ray[0]); r[1] = bk_num((double)ray[1]); r[2] = bk_num((double)ray[2]);
    return 3;
}
