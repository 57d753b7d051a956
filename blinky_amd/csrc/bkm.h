/*
 * bkm.h -- "blinky math": a portable double-precision libm written only with IEEE-754
 * operations (+ - * / sqrt fma, integer bit manipulation).
 *
 * Why it exists.  A Lua lens script evaluates math.sin/cos/atan2/... per output pixel
 * (reference: the Lua 5.2 VM calling the platform libm).  For the GPU lensmap build to be a
 * *function of the script alone*, host and device must agree bit for bit; glibc's and the
 * device library's transcendentals do not.  This header is therefore compiled, unchanged,
 *   - by hiprtc into every generated lensmap-build kernel (device), and
 *   - by the host C++ compiler into the script interpreter (calc_zoom, chunk execution),
 * always with -ffp-contract=off; FMAs appear only where written (__builtin_fma).
 * Same operations in the same order on both sides => identical results by construction.
 *
 * Accuracy: every function is within ~0.5x ulp of the exact value (double-double
 * evaluation of the final steps), i.e. it equals the correctly rounded result except in
 * rare near-tie cases, which is also what glibc returns in all but rare cases
 * (tests/test_bkm.py measures both).  Algorithms are textbook (Cody-Waite / Payne-Hanek
 * reduction, table + Taylor kernels); all constants come from tools/gen_bkm_tables.py.
 */
#ifndef BKM_H
#define BKM_H

#if defined(__HIPCC_RTC__)
#define BKM_FN static __device__ inline
#define BKM_TABLE static __device__ const
typedef unsigned long long bkm_u64;
typedef long long bkm_i64;
typedef unsigned int bkm_u32;
#elif defined(__HIPCC__)
#define BKM_FN static __host__ __device__ inline
#define BKM_TABLE static constexpr
typedef unsigned long long bkm_u64;
typedef long long bkm_i64;
typedef unsigned int bkm_u32;
#else
#define BKM_FN static inline
#define BKM_TABLE static const
typedef unsigned long long bkm_u64;
typedef long long bkm_i64;
typedef unsigned int bkm_u32;
#endif

#include "bkm_tables.h"

typedef struct { double hi, lo; } bkm_dd;

#define BKM_INF (__builtin_inf())
#define BKM_NAN (__builtin_nan(""))

/* ---- bits -------------------------------------------------------------------------- */
BKM_FN bkm_u64 bkm_bits(double x) { bkm_u64 u; __builtin_memcpy(&u, &x, 8); return u; }
BKM_FN double bkm_from_bits(bkm_u64 u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
BKM_FN int bkm_isnan(double x) { return x != x; }
BKM_FN int bkm_isinf(double x) { return __builtin_fabs(x) == BKM_INF; }
BKM_FN double bkm_fabs(double x) { return __builtin_fabs(x); }
BKM_FN double bkm_sqrt(double x) { return __builtin_sqrt(x); }      /* IEEE correctly rounded */
BKM_FN double bkm_copysign(double x, double s) { return __builtin_copysign(x, s); }
BKM_FN double bkm_trunc(double x) { return __builtin_trunc(x); }
BKM_FN double bkm_floor(double x) { return __builtin_floor(x); }
BKM_FN double bkm_ceil(double x) { return __builtin_ceil(x); }

/* round to nearest integer, ties to even, without depending on the rounding-mode libm call:
 * valid for |x| < 2^51 (callers guarantee it) */
BKM_FN double bkm_rint(double x)
{
    const double big = 0x1.8p52;    /* 1.5 * 2^52 */
    return (x + big) - big;
}

/* 2^k for -1022 <= k <= 1023 */
BKM_FN double bkm_pow2(int k) { return bkm_from_bits((bkm_u64)(k + 1023) << 52); }

/* x * 2^k with a single rounding when the result is normal */
BKM_FN double bkm_ldexp(double x, int k)
{
    if (k > 1023) {
        x *= 0x1p1023; k -= 1023;
        if (k > 1023) { x *= 0x1p1023; k -= 1023; if (k > 1023) k = 1023; }
    } else if (k < -1022) {
        x *= 0x1p-969; k += 969;            /* 2^-1022 * 2^53: keep the intermediate normal */
        if (k < -1022) { x *= 0x1p-969; k += 969; if (k < -1022) k = -1022; }
    }
    return x * bkm_pow2(k);
}

/* ---- error-free transformations ---------------------------------------------------- */
BKM_FN bkm_dd bkm_two_sum(double a, double b)
{
    bkm_dd r;
    double bb;
    r.hi = a + b;
    bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}
BKM_FN bkm_dd bkm_fast_two_sum(double a, double b)   /* requires |a| >= |b| or a == 0 */
{
    bkm_dd r;
    r.hi = a + b;
    r.lo = b - (r.hi - a);
    return r;
}
BKM_FN bkm_dd bkm_two_prod(double a, double b)
{
    bkm_dd r;
    r.hi = a * b;
    r.lo = __builtin_fma(a, b, -r.hi);
    return r;
}
/* (ah+al) / (bh+bl) as a double-double, |al|<<|ah|, |bl|<<|bh| */
BKM_FN bkm_dd bkm_dd_div(double ah, double al, double bh, double bl)
{
    bkm_dd q;
    double rem;
    q.hi = ah / bh;
    rem = __builtin_fma(-q.hi, bh, ah);
    q.lo = ((rem + al) - q.hi * bl) / bh;
    return bkm_fast_two_sum(q.hi, q.lo);
}

/* ---- 64x64 -> 128 multiply, portable --------------------------------------------------- */
BKM_FN bkm_u64 bkm_umul128(bkm_u64 a, bkm_u64 b, bkm_u64 *hi)
{
    bkm_u64 a0 = a & 0xFFFFFFFFull, a1 = a >> 32, b0 = b & 0xFFFFFFFFull, b1 = b >> 32;
    bkm_u64 p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    bkm_u64 mid = (p00 >> 32) + (p01 & 0xFFFFFFFFull) + (p10 & 0xFFFFFFFFull);
    *hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    return (p00 & 0xFFFFFFFFull) | (mid << 32);
}

/* ---- argument reduction for sin/cos/tan ---------------------------------------------------
 * returns n (mod 4) and r = x - n*pi/2 as a double-double, |r| <= pi/4 (+1 ulp) */
BKM_FN bkm_u64 bkm_twoopi_bits64(int bitpos)
{
    /* 64 bits of 0.(2/pi) starting at bit `bitpos`, where bit 0 is the first of 64 leading
     * zero bits that precede the binary expansion (so bitpos - 64 indexes bkm_twoopi) */
    int w = (bitpos >> 5) - 2, s = bitpos & 31, i;
    bkm_u32 v[3];
    for (i = 0; i < 3; ++i) {
        int k = w + i;
        v[i] = (k < 0 || k >= 40) ? 0u : bkm_twoopi[k];
    }
    {
        bkm_u64 hi = ((bkm_u64)v[0] << 32) | v[1];
        if (s == 0) return hi;
        return (hi << s) | ((bkm_u64)v[2] >> (32 - s));
    }
}

BKM_FN int bkm_rem_pio2_large(double x, double *rh, double *rl)
{
    /* Payne-Hanek: |x| = M * 2^E, M a 53-bit integer, E >= -32 here */
    bkm_u64 ux = bkm_bits(x) & 0x7FFFFFFFFFFFFFFFull;
    int E = (int)(ux >> 52) - 1075;
    bkm_u64 M = (ux & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull;
    /* window of 192 bits of 2/pi starting at bit (E-1) after the point; +63 = table offset */
    int pos = E + 62;
    bkm_u64 w0 = bkm_twoopi_bits64(pos), w1 = bkm_twoopi_bits64(pos + 64), w2 = bkm_twoopi_bits64(pos + 128);
    /* low 192 bits of M * (w0:w1:w2) */
    bkm_u64 h2, h1, h0, l2, l1, l0, p0, p1, p2, c;
    int n;
    double fh, fl;
    bkm_dd a, b;
    l2 = bkm_umul128(M, w2, &h2);
    l1 = bkm_umul128(M, w1, &h1);
    l0 = bkm_umul128(M, w0, &h0);
    (void)h0;
    p2 = l2;
    p1 = l1 + h2; c = p1 < l1;
    p0 = l0 + h1 + c;
    /* value = (p0:p1:p2) / 2^190 mod 4: top two bits of p0 = quadrant, the rest = fraction */
    n = (int)(p0 >> 62);
    p0 <<= 2; p0 |= p1 >> 62; p1 <<= 2; p1 |= p2 >> 62;
    /* fraction f = (p0:p1) / 2^128 in [0,1); round to nearest quadrant */
    if (p0 >> 63) {
        n = (n + 1) & 3;
        /* f - 1 = -(2^128 - (p0:p1)) / 2^128 */
        p1 = ~p1 + 1; p0 = ~p0 + (p1 == 0);
        fh = -(double)(p0 >> 11) * 0x1p-53;
        fl = -((double)(((p0 & 0x7FF) << 42) | (p1 >> 22))) * 0x1p-106;
    } else {
        fh = (double)(p0 >> 11) * 0x1p-53;
        fl = ((double)(((p0 & 0x7FF) << 42) | (p1 >> 22))) * 0x1p-106;
    }
    /* r = f * pi/2 */
    a = bkm_two_prod(fh, BKM_PIO2_HI);
    a.lo += fh * BKM_PIO2_LO + fl * BKM_PIO2_HI;
    b = bkm_fast_two_sum(a.hi, a.lo);
    *rh = b.hi; *rl = b.lo;
    return n;
}

BKM_FN int bkm_rem_pio2(double x, double *rh, double *rl)
{
    double ax = bkm_fabs(x);
    int n;
    if (ax <= 0x1.921fb54442d18p-1) { *rh = x; *rl = 0.0; return 0; }     /* pi/4 */
    if (ax < 0x1.8p20) {                                                    /* ~1.57e6 */
        double fn = bkm_rint(ax * BKM_2OPI);
        double t = __builtin_fma(-fn, BKM_PIO2_1, ax);                       /* exact */
        bkm_dd s = bkm_two_sum(t, -(fn * BKM_PIO2_2));                       /* fn*P2 exact */
        bkm_dd p3 = bkm_two_prod(fn, BKM_PIO2_3);
        bkm_dd h = bkm_two_sum(s.hi, -p3.hi);
        double lo = h.lo + ((s.lo - p3.lo) - fn * BKM_PIO2_3T);
        bkm_dd r = bkm_fast_two_sum(h.hi, lo);
        n = (int)((bkm_i64)fn & 3);
        if (x < 0) { r.hi = -r.hi; r.lo = -r.lo; n = (4 - n) & 3; }
        *rh = r.hi; *rl = r.lo;
        return n;
    }
    n = bkm_rem_pio2_large(ax, rh, rl);
    if (x < 0) { *rh = -*rh; *rl = -*rl; n = (4 - n) & 3; }
    return n;
}

/* Second-stage reduction and evaluation.  With x = n*(pi/2) + r (bkm_rem_pio2) write
 * r = j*(pi/32) + t, |t| <= pi/64, so that x = m*(pi/32) + t with m = 16 n + j (mod 64) and
 *   sin(x) = S + C*t + [ S*(cos t - 1) + C*(sin t - t) ],   S = sin(m pi/32), C = cos(m pi/32)
 * where S, C come from a table as hi+lo pairs and the bracket is <= 1.3e-3 |result|: its
 * rounding errors are invisible, the dominant C*t is formed exactly (two_prod). */
BKM_FN int bkm_rem_pio32(double x, double *th, double *tl)
{
    double rh, rl, fj, a;
    int n = bkm_rem_pio2(x, &rh, &rl), j;
    bkm_dd s, p, h, r;
    fj = bkm_rint(rh * BKM_32OPI);                  /* |fj| <= 8 */
    j = (int)fj;
    a = __builtin_fma(-fj, BKM_PIO32_1, rh);        /* exact */
    p = bkm_two_prod(fj, BKM_PIO32_2);
    s = bkm_two_sum(a, -p.hi);
    h = bkm_two_sum(s.hi, rl);
    r = bkm_fast_two_sum(h.hi, h.lo + ((s.lo - p.lo) - fj * BKM_PIO32_2T));
    *th = r.hi; *tl = r.lo;
    return (16 * n + j) & 63;
}

/* sin(m*pi/32 + t) as a double-double */
BKM_FN bkm_dd bkm_sin_mt(int m, double th, double tl)
{
    double Sh = bkm_sin_tab[m][0], Sl = bkm_sin_tab[m][1];
    double Ch = bkm_sin_tab[(m + 16) & 63][0], Cl = bkm_sin_tab[(m + 16) & 63][1];
    double z = th * th, sp, cp, rest;
    bkm_dd p, s;
    sp = bkm_sin_c[5];
    sp = sp * z + bkm_sin_c[4];
    sp = sp * z + bkm_sin_c[3];
    sp = sp * z + bkm_sin_c[2];
    sp = sp * z + bkm_sin_c[1];
    sp = sp * z + bkm_sin_c[0];
    sp = (th * z) * sp;                              /* sin t - t */
    cp = bkm_cos_c[5];
    cp = cp * z + bkm_cos_c[4];
    cp = cp * z + bkm_cos_c[3];
    cp = cp * z + bkm_cos_c[2];
    cp = cp * z + bkm_cos_c[1];
    cp = cp * z + bkm_cos_c[0];
    cp = (z * z) * cp - 0.5 * z;                     /* cos t - 1 */
    rest = ((Sh * cp + Ch * sp) + (Ch * tl + Cl * th)) + Sl;
    p = bkm_two_prod(Ch, th);
    s = bkm_two_sum(Sh, p.hi);
    return bkm_fast_two_sum(s.hi, s.lo + (p.lo + rest));
}

BKM_FN double bkm_sin(double x)
{
    double th, tl;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-26) return x;
    m = bkm_rem_pio32(x, &th, &tl);
    return bkm_sin_mt(m, th, tl).hi;
}
BKM_FN double bkm_cos(double x)
{
    double th, tl;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-27) return 1.0;
    m = bkm_rem_pio32(x, &th, &tl);
    return bkm_sin_mt((m + 16) & 63, th, tl).hi;
}
BKM_FN double bkm_tan(double x)
{
    double th, tl;
    bkm_dd s, c, q;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-27) return x;
    m = bkm_rem_pio32(x, &th, &tl);
    s = bkm_sin_mt(m, th, tl);
    c = bkm_sin_mt((m + 16) & 63, th, tl);
    q = bkm_dd_div(s.hi, s.lo, c.hi, c.lo);
    return q.hi;
}

/* ---- atan family -------------------------------------------------------------------------- */
/* atan(qh+ql) for qh >= 0 (may be +inf), as a double-double */
BKM_FN bkm_dd bkm_atan_dd(double qh, double ql)
{
    double uh = qh, ul = ql, th, tl, z, a, pl;
    bkm_dd r, s;
    int inv = 0, i;
    if (qh == BKM_INF) { r.hi = BKM_PIO2_HI; r.lo = BKM_PIO2_LO; return r; }
    if (qh > 1.0) {
        double e;
        inv = 1;
        uh = 1.0 / qh;
        e = __builtin_fma(-uh, qh, 1.0);
        ul = (e - uh * ql) / qh;
    }
    i = (int)bkm_rint(uh * 8.0);
    if (i == 0) {
        th = uh; tl = ul;
    } else {
        double c = (double)i * 0.125;
        double nh = uh - c;                         /* exact (Sterbenz) */
        bkm_dd d = bkm_two_sum(1.0, uh * c);        /* uh*c exact: c has <= 4 bits */
        bkm_dd t;
        d.lo += ul * c;
        t = bkm_dd_div(nh, ul, d.hi, d.lo);
        th = t.hi; tl = t.lo;
    }
    z = th * th;
    a = bkm_atan_c[8];
    a = a * z + bkm_atan_c[7];
    a = a * z + bkm_atan_c[6];
    a = a * z + bkm_atan_c[5];
    a = a * z + bkm_atan_c[4];
    a = a * z + bkm_atan_c[3];
    a = a * z + bkm_atan_c[2];
    a = a * z + bkm_atan_c[1];
    a = a * z + bkm_atan_c[0];
    pl = (th * z) * a;
    s = bkm_two_sum(bkm_atan_tab[i][0], th);
    r = bkm_fast_two_sum(s.hi, s.lo + ((bkm_atan_tab[i][1] + tl) + pl));
    if (inv) {
        s = bkm_two_sum(BKM_PIO2_HI, -r.hi);
        r = bkm_fast_two_sum(s.hi, s.lo + (BKM_PIO2_LO - r.lo));
    }
    return r;
}

BKM_FN double bkm_atan(double x)
{
    bkm_dd r;
    if (bkm_isnan(x)) return x;
    if (bkm_fabs(x) < 0x1p-27) return x;
    r = bkm_atan_dd(bkm_fabs(x), 0.0);
    return bkm_copysign(r.hi, x);
}

BKM_FN double bkm_atan2(double y, double x)
{
    double ax, ay;
    bkm_dd q, a;
    if (bkm_isnan(x) || bkm_isnan(y)) return BKM_NAN;
    ax = bkm_fabs(x); ay = bkm_fabs(y);
    if (ay == 0.0) {                                            /* y = +-0 */
        if (bkm_bits(x) >> 63) return bkm_copysign(BKM_PI_HI, y);   /* x < 0 or -0 */
        return y;
    }
    if (ax == 0.0) return bkm_copysign(BKM_PIO2_HI, y);
    if (ax == BKM_INF) {
        if (ay == BKM_INF) return bkm_copysign((bkm_bits(x) >> 63) ? 3.0 * BKM_PIO4_HI : BKM_PIO4_HI, y);
        return (bkm_bits(x) >> 63) ? bkm_copysign(BKM_PI_HI, y) : bkm_copysign(0.0, y);
    }
    if (ay == BKM_INF) return bkm_copysign(BKM_PIO2_HI, y);
    /* q = ay/ax as a double-double (scaled away from overflow/underflow of the remainder) */
    {
        int ex = (int)((bkm_bits(ax) >> 52) & 0x7FF), ey = (int)((bkm_bits(ay) >> 52) & 0x7FF);
        if (ey - ex > 60) {                                     /* |y/x| > 2^59: pi/2 to full precision */
            a.hi = BKM_PIO2_HI; a.lo = BKM_PIO2_LO;
            if (ey - ex < 200) {                                /* first-order correction -x/y */
                double c = ax / ay;
                a = bkm_fast_two_sum(a.hi, a.lo - c);
            }
        } else if (ex - ey > 1000 || ex == 0 || ey == 0 || ex > 2000 || ey > 2000) {
            /* tiny quotient or subnormal/huge operands: rescale both into the normal range */
            double sx = ax, sy = ay;
            if (ex > 1500 || ey > 1500) { sx *= 0x1p-600; sy *= 0x1p-600; }
            else if (ex < 500 && ey < 500) { sx *= 0x1p600; sy *= 0x1p600; }
            q.hi = sy / sx;
            if (q.hi < 0x1p-900) {                              /* atan(q) = q to full precision */
                if (bkm_bits(x) >> 63) return bkm_copysign(BKM_PI_HI, y);
                return bkm_copysign(ay / ax, y);
            }
            q.lo = __builtin_fma(-q.hi, sx, sy) / sx;
            a = bkm_atan_dd(q.hi, q.lo);
        } else {
            q.hi = ay / ax;
            q.lo = __builtin_fma(-q.hi, ax, ay) / ax;
            a = bkm_atan_dd(q.hi, q.lo);
        }
    }
    if (bkm_bits(x) >> 63) {                                    /* x < 0: pi - a */
        bkm_dd s = bkm_two_sum(BKM_PI_HI, -a.hi);
        a = bkm_fast_two_sum(s.hi, s.lo + (BKM_PI_LO - a.lo));
    }
    return bkm_copysign(a.hi, y);
}

/* sqrt((1-ax)(1+ax)) as a double-double, 0 <= ax <= 1 */
BKM_FN bkm_dd bkm_sqrt1mx2(double ax)
{
    bkm_dd a = bkm_two_sum(1.0, -ax), b = bkm_two_sum(1.0, ax), p, s;
    p = bkm_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    s.hi = bkm_sqrt(p.hi);
    if (s.hi == 0.0) { s.lo = 0.0; return s; }
    s.lo = (__builtin_fma(-s.hi, s.hi, p.hi) + p.lo) / (2.0 * s.hi);
    return s;
}

BKM_FN double bkm_asin(double x)
{
    double ax = bkm_fabs(x);
    bkm_dd s, q, a;
    if (bkm_isnan(x)) return x;
    if (ax > 1.0) return BKM_NAN;
    if (ax < 0x1p-27) return x;
    s = bkm_sqrt1mx2(ax);
    if (s.hi == 0.0) return bkm_copysign(BKM_PIO2_HI, x);
    q = bkm_dd_div(ax, 0.0, s.hi, s.lo);
    a = bkm_atan_dd(q.hi, q.lo);
    return bkm_copysign(a.hi, x);
}

BKM_FN double bkm_acos(double x)
{
    double ax = bkm_fabs(x);
    bkm_dd s, q, a;
    if (bkm_isnan(x)) return x;
    if (ax > 1.0) return BKM_NAN;
    if (x == 1.0) return 0.0;
    if (ax < 0x1p-60) return BKM_PIO2_HI;
    s = bkm_sqrt1mx2(ax);
    q = bkm_dd_div(s.hi, s.lo, ax, 0.0);          /* s/|x|, may be large; s = 0 -> 0 */
    a = bkm_atan_dd(q.hi, q.lo);
    if (x < 0) {
        bkm_dd t = bkm_two_sum(BKM_PI_HI, -a.hi);
        a = bkm_fast_two_sum(t.hi, t.lo + (BKM_PI_LO - a.lo));
    }
    return a.hi;
}

/* ---- exp family ----------------------------------------------------------------------------- */
/* e^(xh+xl) = 2^k * (m.hi + m.lo), |xh| < 746, m in [1,2) */
BKM_FN bkm_dd bkm_exp_dd(double xh, double xl, int *k)
{
    double kf = bkm_rint(xh * BKM_64OLN2);
    double rh = __builtin_fma(-kf, BKM_LN2O64_HI, xh);           /* exact */
    double rl = __builtin_fma(-kf, BKM_LN2O64_LO, xl);
    bkm_dd r = bkm_two_sum(rh, rl);
    int ki = (int)kf, j = ki & 63;
    double th = bkm_exp_tab[j][0], tl = bkm_exp_tab[j][1];
    double e, pl;
    bkm_dd a, s;
    *k = (ki - j) / 64;
    e = bkm_exp_c[6];
    e = e * r.hi + bkm_exp_c[5];
    e = e * r.hi + bkm_exp_c[4];
    e = e * r.hi + bkm_exp_c[3];
    e = e * r.hi + bkm_exp_c[2];
    e = e * r.hi + bkm_exp_c[1];
    e = e * r.hi + bkm_exp_c[0];
    pl = r.lo + (r.hi * r.hi) * e;                               /* expm1(r) = r.hi + pl */
    a = bkm_two_prod(th, r.hi);
    s = bkm_two_sum(th, a.hi);
    return bkm_fast_two_sum(s.hi, s.lo + (a.lo + (th * pl + (tl + tl * r.hi))));
}

/* (m.hi + m.lo) * 2^k rounded once, including results in the subnormal range */
BKM_FN double bkm_scale_dd(bkm_dd m, int k)
{
    if (k >= -1021) return bkm_ldexp(m.hi, k);
    if (k < -1080) return 0.0 * m.hi;
    {
        /* u = m * 2^(k+1022) in (0,2): adding 1.0 rounds it on the 2^-52 grid = the subnormal grid */
        double sc = bkm_pow2(k + 1022 > -1022 ? k + 1022 : -1022);
        double u = m.hi * sc, ul = m.lo * sc;
        bkm_dd t = bkm_two_sum(1.0, u);
        double w = t.hi + (t.lo + ul);
        return (w - 1.0) * 0x1p-1022;
    }
}

BKM_FN double bkm_exp(double x)
{
    bkm_dd m;
    int k;
    if (bkm_isnan(x)) return x;
    if (x > 709.782712893384) return BKM_INF;
    if (x < -745.1332191019412) return 0.0;
    if (bkm_fabs(x) < 0x1p-54) return 1.0 + x;
    m = bkm_exp_dd(x, 0.0, &k);
    return bkm_scale_dd(m, k);
}

/* sinh as a double-double for 0 <= x < 0.35 (Taylor), hi part exact-ish */
BKM_FN bkm_dd bkm_sinh_small(double x)
{
    double z = x * x, h;
    h = bkm_sinh_c[7];
    h = h * z + bkm_sinh_c[6];
    h = h * z + bkm_sinh_c[5];
    h = h * z + bkm_sinh_c[4];
    h = h * z + bkm_sinh_c[3];
    h = h * z + bkm_sinh_c[2];
    h = h * z + bkm_sinh_c[1];
    return bkm_fast_two_sum(x, (x * z) * (h * z) + (x * z) * bkm_sinh_c[0]);
}

/* (e^x -+ e^-x)/2 for 0.35 <= x < 40 as double-doubles */
BKM_FN void bkm_sinhcosh_dd(double x, bkm_dd *sh, bkm_dd *ch)
{
    int k;
    bkm_dd m = bkm_exp_dd(x, 0.0, &k), inv, s, a, b;
    double sc = bkm_pow2(k - 1), isc = bkm_pow2(-k - 1);
    inv.hi = 1.0 / m.hi;
    inv.lo = (__builtin_fma(-inv.hi, m.hi, 1.0) - inv.hi * m.lo) / m.hi;
    a.hi = m.hi * sc; a.lo = m.lo * sc;                         /* e^x / 2 */
    b.hi = inv.hi * isc; b.lo = inv.lo * isc;                   /* e^-x / 2 */
    s = bkm_two_sum(a.hi, -b.hi);
    *sh = bkm_fast_two_sum(s.hi, s.lo + (a.lo - b.lo));
    s = bkm_two_sum(a.hi, b.hi);
    *ch = bkm_fast_two_sum(s.hi, s.lo + (a.lo + b.lo));
}

BKM_FN double bkm_sinh(double x)
{
    double ax = bkm_fabs(x);
    bkm_dd s, c;
    if (bkm_isnan(x) || bkm_isinf(x)) return x;
    if (ax < 0x1p-28) return x;
    if (ax < 0.35) { s = bkm_sinh_small(ax); return bkm_copysign(s.hi, x); }
    if (ax >= 40.0) {
        int k;
        if (ax > 710.4758600739439) return bkm_copysign(BKM_INF, x);
        s = bkm_exp_dd(ax, 0.0, &k);
        return bkm_copysign(bkm_ldexp(s.hi, k - 1), x);
    }
    bkm_sinhcosh_dd(ax, &s, &c);
    return bkm_copysign(s.hi, x);
}
BKM_FN double bkm_cosh(double x)
{
    double ax = bkm_fabs(x);
    bkm_dd s, c;
    if (bkm_isnan(x)) return x;
    if (ax == BKM_INF) return BKM_INF;
    if (ax < 0x1p-28) return 1.0;
    if (ax >= 40.0) {
        int k;
        if (ax > 710.4758600739439) return BKM_INF;
        s = bkm_exp_dd(ax, 0.0, &k);
        return bkm_ldexp(s.hi, k - 1);
    }
    if (ax < 0.35) {
        /* cosh = 1 + 2 sinh^2(x/2) */
        bkm_dd h = bkm_sinh_small(0.5 * ax), p = bkm_two_prod(h.hi, h.hi), t;
        p.lo += 2.0 * h.hi * h.lo;
        t = bkm_two_sum(1.0, 2.0 * p.hi);
        return t.hi + (t.lo + 2.0 * p.lo);
    }
    bkm_sinhcosh_dd(ax, &s, &c);
    return c.hi;
}
BKM_FN double bkm_tanh(double x)
{
    double ax = bkm_fabs(x);
    bkm_dd s, c, q;
    if (bkm_isnan(x)) return x;
    if (ax < 0x1p-28) return x;
    if (ax >= 22.0) return bkm_copysign(1.0, x);
    if (ax < 0.35) {
        bkm_dd h = bkm_sinh_small(0.5 * ax), p = bkm_two_prod(h.hi, h.hi), t;
        s = bkm_sinh_small(ax);
        p.lo += 2.0 * h.hi * h.lo;
        t = bkm_two_sum(1.0, 2.0 * p.hi);
        c = bkm_fast_two_sum(t.hi, t.lo + 2.0 * p.lo);
    } else {
        bkm_sinhcosh_dd(ax, &s, &c);
    }
    q = bkm_dd_div(s.hi, s.lo, c.hi, c.lo);
    return bkm_copysign(q.hi, x);
}

/* ---- log family ----------------------------------------------------------------------------- */
/* log(x) for finite x > 0 as a double-double */
BKM_FN bkm_dd bkm_log_dd(double x)
{
    bkm_u64 u = bkm_bits(x);
    int k = 0, i;
    double m, rc, q, l;
    bkm_dd p, r, s, t;
    if ((u >> 52) == 0) { x *= 0x1p54; k = -54; u = bkm_bits(x); }     /* subnormal */
    k += (int)(u >> 52) - 1023;
    m = bkm_from_bits((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);  /* [1,2) */
    i = (int)(((u >> 44) & 0xFF) + 1) >> 1;                              /* round((m-1)*128) */
    if (i == 128) { k += 1; m *= 0.5; i = 0; }      /* m ~ 2: log(m/2) + (k+1) ln2, no cancellation at x ~ 1 */
    rc = bkm_log_tab[i][0];
    p = bkm_two_prod(m, rc);
    r = bkm_two_sum(p.hi - 1.0, p.lo);                                   /* r = m*rc - 1, exact */
    l = bkm_log_c[7];
    l = l * r.hi + bkm_log_c[6];
    l = l * r.hi + bkm_log_c[5];
    l = l * r.hi + bkm_log_c[4];
    l = l * r.hi + bkm_log_c[3];
    l = l * r.hi + bkm_log_c[2];
    l = l * r.hi + bkm_log_c[1];
    l = l * r.hi + bkm_log_c[0];
    q = (r.hi * r.hi) * l - r.hi * r.lo;                                 /* log1p(r) - r */
    /* k*ln2 + (-log rc) + r */
    s = bkm_two_sum((double)k * BKM_LN2_HI, bkm_log_tab[i][1]);          /* k*LN2_HI exact */
    t = bkm_two_sum(s.hi, r.hi);
    return bkm_fast_two_sum(t.hi, t.lo + (s.lo + ((double)k * BKM_LN2_LO + bkm_log_tab[i][2] + r.lo + q)));
}

BKM_FN double bkm_log(double x)
{
    bkm_dd l;
    if (bkm_isnan(x)) return x;
    if (x == 0.0) return -BKM_INF;
    if (x < 0.0) return BKM_NAN;
    if (x == BKM_INF) return x;
    if (x == 1.0) return 0.0;
    l = bkm_log_dd(x);
    return l.hi;
}
BKM_FN double bkm_log10(double x)
{
    bkm_dd l, p;
    if (bkm_isnan(x)) return x;
    if (x == 0.0) return -BKM_INF;
    if (x < 0.0) return BKM_NAN;
    if (x == BKM_INF) return x;
    if (x == 1.0) return 0.0;
    l = bkm_log_dd(x);
    p = bkm_two_prod(l.hi, BKM_INVLN10_HI);
    return p.hi + (p.lo + (l.hi * BKM_INVLN10_LO + l.lo * BKM_INVLN10_HI));
}

/* is y an integer? returns 0 no, 1 odd integer, 2 even integer */
BKM_FN int bkm_intclass(double y)
{
    double ay = bkm_fabs(y), t;
    if (ay >= 0x1p53) return 2;
    t = bkm_trunc(ay);
    if (t != ay) return 0;
    return ((bkm_u64)t & 1) ? 1 : 2;
}

BKM_FN double bkm_pow(double x, double y)
{
    double ax, sign = 1.0;
    bkm_dd l, p, m;
    int k, yi;
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (bkm_isnan(x) || bkm_isnan(y)) return BKM_NAN;
    yi = bkm_intclass(y);
    ax = bkm_fabs(x);
    if (bkm_isinf(y)) {
        if (ax == 1.0) return 1.0;
        if ((ax > 1.0) == (y > 0)) return BKM_INF;
        return 0.0;
    }
    if (ax == 0.0 || ax == BKM_INF) {
        double r = ((ax == 0.0) == (y > 0)) ? 0.0 : BKM_INF;
        if (yi == 1 && (bkm_bits(x) >> 63)) r = -r;
        return r;
    }
    if (x < 0) {
        if (!yi) return BKM_NAN;
        if (yi == 1) sign = -1.0;
    }
    if (y == 2.0) return x * x;            /* exact rounding, and what any good pow returns */
    if (y == 1.0) return x;
    if (y == -1.0) return 1.0 / x;
    if (y == 0.5 && x >= 0) return bkm_sqrt(x);
    l = bkm_log_dd(ax);
    p = bkm_two_prod(y, l.hi);
    p.lo += y * l.lo;
    p = bkm_fast_two_sum(p.hi, p.lo);
    if (p.hi > 709.79) return sign * BKM_INF;
    if (p.hi < -745.2) return sign * 0.0;
    m = bkm_exp_dd(p.hi, p.lo, &k);
    return sign * bkm_scale_dd(m, k);
}

/* ---- fmod: exact, by shift-and-subtract on the integer significands ------------------------------ */
BKM_FN double bkm_fmod(double x, double y)
{
    bkm_u64 ux = bkm_bits(x), uy = bkm_bits(y), sx = ux & 0x8000000000000000ull, mx, my;
    int ex, ey;
    ux &= 0x7FFFFFFFFFFFFFFFull; uy &= 0x7FFFFFFFFFFFFFFFull;
    if (uy == 0 || ux >= 0x7FF0000000000000ull || uy > 0x7FF0000000000000ull) return BKM_NAN;
    if (ux < uy) return x;
    if (ux == uy) return bkm_from_bits(sx);              /* +-0 */
    ex = (int)(ux >> 52); ey = (int)(uy >> 52);
    mx = ux & 0x000FFFFFFFFFFFFFull; my = uy & 0x000FFFFFFFFFFFFFull;
    if (ex == 0) { ex = 1; while (!(mx >> 52)) { mx <<= 1; --ex; } } else mx |= 1ull << 52;
    if (ey == 0) { ey = 1; while (!(my >> 52)) { my <<= 1; --ey; } } else my |= 1ull << 52;
    for (; ex > ey; --ex) {
        if (mx >= my) mx -= my;
        mx <<= 1;
    }
    if (mx >= my) mx -= my;
    if (mx == 0) return bkm_from_bits(sx);
    while (!(mx >> 52)) { mx <<= 1; --ex; }
    if (ex > 0) ux = (mx & 0x000FFFFFFFFFFFFFull) | ((bkm_u64)ex << 52);
    else ux = mx >> (1 - ex);
    return bkm_from_bits(ux | sx);
}

#endif /* BKM_H */
