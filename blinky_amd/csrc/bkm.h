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

/* One Horner step a * z + c with a coefficient c.  On the device the coefficient is handed to v_fma_f64 as an SGPR pair: left to itself
 * the compiler picks v_fmac_f64, whose addend is its destination, and pays two v_mov_b32 per coefficient to get the literal there. */
#if defined(__HIPCC_RTC__)
static __device__ inline double bkm_horner(double a, double z, double c)
{
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(z), "s"(c));
    return r;
}
#else
#define bkm_horner(a, z, c) __builtin_fma((a), (z), (c))
#endif

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
/* (ah+al) / (bh+bl) as a double-double, |al|<<|ah|, |bl|<<|bh|: ONE division (a reciprocal); the quotient's head need not be
 * correctly rounded, its tail makes up for it */
BKM_FN bkm_dd bkm_dd_div(double ah, double al, double bh, double bl)
{
    bkm_dd q;
    double rem, r = 1.0 / bh;
    q.hi = ah * r;
    rem = __builtin_fma(-q.hi, bh, ah);
    q.lo = ((rem + al) - q.hi * bl) * r;
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

/* Reduction and evaluation.  Write x = m*(pi/32) + t, |t| <= pi/64, m mod 64; then
 *   sin(x) = S + C*t + [ S*(cos t - 1) + C*(sin t - t) ],   S = sin(m pi/32), C = cos(m pi/32)
 * where S, C come from a table as hi+lo pairs and the bracket is <= 1.3e-3 |result|: its
 * rounding errors are invisible, the dominant C*t is formed exactly (two_prod).
 *
 * |x| < 2^16 - every argument a lens script produces - takes ONE Cody-Waite step with pi/32 = P1 + P2 + P2T (33 + 53 + 53
 * bits): fn = rint(x * 32/pi) has at most 20 bits, so fn*P1 is exact; x - fn*P1 is exact as well (for fn != 0 both lie on a
 * grid no finer than 2^-57 and the difference is below 2^-4); what is left of pi/32 after 139 bits is below 2^-143, i.e. the
 * reduced argument is off by < 2^-123 where no double in range comes closer than 2^-70 to a multiple of pi/32.  Larger
 * arguments go through pi/2 first (Cody-Waite to 2^20.6, Payne-Hanek beyond), then pi/32. */
BKM_FN int bkm_rem_pio32(double x, double *th, double *tl)
{
    double rh, rl, fj, a;
    int n, j;
    bkm_dd s, p, h, r;
    if (bkm_fabs(x) < 0x1p16) {
        fj = bkm_rint(x * BKM_32OPI);
        a = __builtin_fma(-fj, BKM_PIO32_1, x);     /* exact */
        p = bkm_two_prod(fj, BKM_PIO32_2);
        s = bkm_two_sum(a, -p.hi);
        r = bkm_fast_two_sum(s.hi, (s.lo - p.lo) - fj * BKM_PIO32_2T);
        *th = r.hi; *tl = r.lo;
        return (int)fj & 63;
    }
    n = bkm_rem_pio2(x, &rh, &rl);
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

/* sin t - t and cos t - 1 for |t| <= pi/64 (+ an ulp): Taylor through t^9 / t^10; what is dropped is below 2^-68 of t and
 * 2^-80 of 1.  One pair serves sin AND cos of the same argument. */
typedef struct { double sp, cp; } bkm_scp;
BKM_FN bkm_scp bkm_sincos_poly(double th)
{
    bkm_scp q;
    double z = th * th, sp, cp;
    sp = bkm_sin_c[3];
    sp = bkm_horner(sp, z, bkm_sin_c[2]);
    sp = bkm_horner(sp, z, bkm_sin_c[1]);
    sp = bkm_horner(sp, z, bkm_sin_c[0]);
    q.sp = (th * z) * sp;
    cp = bkm_cos_c[3];
    cp = bkm_horner(cp, z, bkm_cos_c[2]);
    cp = bkm_horner(cp, z, bkm_cos_c[1]);
    cp = bkm_horner(cp, z, bkm_cos_c[0]);
    q.cp = __builtin_fma(z * z, cp, -0.5 * z);
    return q;
}

/* sin(m*pi/32 + t) as a double-double */
BKM_FN bkm_dd bkm_sin_mt(int m, double th, double tl, bkm_scp q)
{
    double Sh = bkm_sin_tab[m][0], Sl = bkm_sin_tab[m][1];
    double Ch = bkm_sin_tab[(m + 16) & 63][0], Cl = bkm_sin_tab[(m + 16) & 63][1];
    double rest = (__builtin_fma(Sh, q.cp, Ch * q.sp) + __builtin_fma(Ch, tl, Cl * th)) + Sl;
    bkm_dd p = bkm_two_prod(Ch, th);
    bkm_dd s = bkm_fast_two_sum(Sh, p.hi);           /* S = 0 exactly (m = 0, 32) or |S| >= sin(pi/32) > |C t| */
    return bkm_fast_two_sum(s.hi, s.lo + (p.lo + rest));
}

BKM_FN double bkm_sin(double x)
{
    double th, tl;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-26) return x;
    m = bkm_rem_pio32(x, &th, &tl);
    return bkm_sin_mt(m, th, tl, bkm_sincos_poly(th)).hi;
}
BKM_FN double bkm_cos(double x)
{
    double th, tl;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-27) return 1.0;
    m = bkm_rem_pio32(x, &th, &tl);
    return bkm_sin_mt((m + 16) & 63, th, tl, bkm_sincos_poly(th)).hi;
}
/* both at once: one reduction, one pair of polynomials; *s == bkm_sin(x) and *c == bkm_cos(x) bit for bit */
BKM_FN void bkm_sincos(double x, double *s, double *c)
{
    double th, tl, ax = bkm_fabs(x);
    bkm_scp q;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) { *s = BKM_NAN; *c = BKM_NAN; return; }
    m = bkm_rem_pio32(x, &th, &tl);
    q = bkm_sincos_poly(th);
    *s = ax < 0x1p-26 ? x : bkm_sin_mt(m, th, tl, q).hi;
    *c = ax < 0x1p-27 ? 1.0 : bkm_sin_mt((m + 16) & 63, th, tl, q).hi;
}
BKM_FN double bkm_tan(double x)
{
    double th, tl;
    bkm_dd s, c, q;
    bkm_scp pq;
    int m;
    if (bkm_isnan(x) || bkm_isinf(x)) return BKM_NAN;
    if (bkm_fabs(x) < 0x1p-27) return x;
    m = bkm_rem_pio32(x, &th, &tl);
    pq = bkm_sincos_poly(th);
    s = bkm_sin_mt(m, th, tl, pq);
    c = bkm_sin_mt((m + 16) & 63, th, tl, pq);
    q = bkm_dd_div(s.hi, s.lo, c.hi, c.lo);
    return q.hi;
}

/* ---- atan family -------------------------------------------------------------------------- */
/* atan(N/D) for 0 <= N <= D given as double-doubles (nh <= dh, dh finite and normal), as an UNNORMALISED hi + lo.
 * u = N/D is never formed: with c = i/8 the table entry nearest to u,
 *   atan(N/D) = atan(c) + atan(t),   t = (N - c D) / (D + c N),   |t| <= 1/16 (+ 2^-12)
 * costs ONE division, the reciprocal of D + c N.  i only has to be near 8u, so it comes from a rough 1/dh (exponent trick +
 * two Newton steps: 7e-6) - biased down by 2^-10 so that i >= 1 implies nh >= dh/16, which makes nh - c dh exact.
 * atan t - t: Taylor through t^15; what is dropped is below 2^-68 of t. */
BKM_FN bkm_dd bkm_atan_frac(double nh, double nl, double dh, double dl)
{
    double r, e, c, Nh, Nl, rD, th, tl, rem, z, a, pl;
    bkm_dd cn, D, s, o;
    int i;
    r = bkm_from_bits(0x7FDE623822FC16E6ull - bkm_bits(dh));
    e = __builtin_fma(-dh, r, 1.0); r = __builtin_fma(r, e, r);
    e = __builtin_fma(-dh, r, 1.0); r = __builtin_fma(r, e, r);
    i = (int)bkm_rint(__builtin_fma(nh * r, 8.0, -0x1p-10));    /* 0 .. 8 */
    c = (double)i * 0.125;
    Nh = __builtin_fma(-c, dh, nh);                            /* exact */
    Nl = __builtin_fma(-c, dl, nl);
    cn = bkm_two_prod(c, nh);
    D = bkm_fast_two_sum(dh, cn.hi);                           /* dh >= c nh */
    D.lo += cn.lo + __builtin_fma(c, nl, dl);
    rD = 1.0 / D.hi;
    th = Nh * rD;
    rem = __builtin_fma(-th, D.hi, Nh);
    tl = ((rem + Nl) - th * D.lo) * rD;
    z = th * th;
    a = bkm_atan_c[6];
    a = bkm_horner(a, z, bkm_atan_c[5]);
    a = bkm_horner(a, z, bkm_atan_c[4]);
    a = bkm_horner(a, z, bkm_atan_c[3]);
    a = bkm_horner(a, z, bkm_atan_c[2]);
    a = bkm_horner(a, z, bkm_atan_c[1]);
    a = bkm_horner(a, z, bkm_atan_c[0]);
    pl = (th * z) * a;
    s = bkm_fast_two_sum(bkm_atan_tab[i][0], th);              /* entry 0 is 0; the others are >= 0.124 > |t| */
    o.hi = s.hi;
    o.lo = s.lo + ((bkm_atan_tab[i][1] + tl) + pl);
    return o;
}

/* the angle of the point (+-X, Y), X, Y >= 0, from r = atan(min/max) (bkm_atan_frac): swapped = Y was the larger one */
BKM_FN double bkm_atan_place(bkm_dd r, int swapped, int xneg)
{
    double kh = 0.0, kl = 0.0, sg = 1.0;
    bkm_dd f;
    if (swapped) { kh = BKM_PIO2_HI; kl = BKM_PIO2_LO; sg = xneg ? 1.0 : -1.0; }      /* pi/2 -+ r */
    else if (xneg) { kh = BKM_PI_HI; kl = BKM_PI_LO; sg = -1.0; }                     /* pi - r */
    f = bkm_fast_two_sum(kh, sg * r.hi);                       /* k = 0 or k >= pi/2 > r */
    return f.hi + (f.lo + (kl + sg * r.lo));
}

BKM_FN double bkm_atan2(double y, double x);
BKM_FN double bkm_atan(double x)
{
    if (bkm_isnan(x)) return x;
    if (bkm_fabs(x) < 0x1p-27) return x;
    return bkm_atan2(x, 1.0);
}

BKM_FN double bkm_atan2(double y, double x)
{
    double ax, ay, num, den;
    bkm_dd r;
    int swapped, en, ed;
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
    swapped = ay > ax;
    num = swapped ? ax : ay; den = swapped ? ay : ax;
    en = (int)(bkm_bits(num) >> 52); ed = (int)(bkm_bits(den) >> 52);
    if (ed - en > 60) {
        /* the smaller one is below 2^-59 of the larger: atan q = q to full precision (q may be subnormal or 0: so is the result) */
        r.hi = num / den; r.lo = 0.0;
    } else {
        if (ed > 1900) { num *= 0x1p-600; den *= 0x1p-600; }    /* exact: num >= 2^-61 den */
        else if (ed < 200) { num *= 0x1p600; den *= 0x1p600; }  /* exact, and both normal afterwards */
        r = bkm_atan_frac(num, 0.0, den, 0.0);
    }
    return bkm_copysign(bkm_atan_place(r, swapped, (int)(bkm_bits(x) >> 63)), y);
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

BKM_FN double bkm_asin(double x)                   /* atan2(|x|, sqrt(1 - x^2)) */
{
    double ax = bkm_fabs(x);
    bkm_dd s, r;
    if (bkm_isnan(x)) return x;
    if (ax > 1.0) return BKM_NAN;
    if (ax < 0x1p-27) return x;
    s = bkm_sqrt1mx2(ax);
    if (s.hi == 0.0) return bkm_copysign(BKM_PIO2_HI, x);
    r = ax > s.hi ? bkm_atan_frac(s.hi, s.lo, ax, 0.0) : bkm_atan_frac(ax, 0.0, s.hi, s.lo);
    return bkm_copysign(bkm_atan_place(r, ax > s.hi, 0), x);
}

BKM_FN double bkm_acos(double x)                   /* atan2(sqrt(1 - x^2), x) */
{
    double ax = bkm_fabs(x);
    bkm_dd s, r;
    if (bkm_isnan(x)) return x;
    if (ax > 1.0) return BKM_NAN;
    if (x == 1.0) return 0.0;
    if (ax < 0x1p-60) return BKM_PIO2_HI;
    s = bkm_sqrt1mx2(ax);
    r = s.hi > ax ? bkm_atan_frac(ax, 0.0, s.hi, s.lo) : bkm_atan_frac(s.hi, s.lo, ax, 0.0);
    return bkm_atan_place(r, s.hi > ax, x < 0);
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

/* ---- fmod: exact ------------------------------------------------------------------------------------
 * Quotients below 2^52 (every call a lens script or the rubix grid makes): q = trunc(|x| / |y|) is the true integer quotient or,
 * when the division rounded up across an integer, one more; x - q y is exactly representable either way (a multiple of ulp(y) of
 * magnitude <= |y|), so ONE fma gives it exactly and a negative result is put right by adding |y|, exactly again.  Anything else:
 * shift-and-subtract on the integer significands. */
BKM_FN double bkm_fmod(double x, double y)
{
    bkm_u64 ux = bkm_bits(x), uy = bkm_bits(y), sx = ux & 0x8000000000000000ull, mx, my;
    int ex, ey, sh;
    ux &= 0x7FFFFFFFFFFFFFFFull; uy &= 0x7FFFFFFFFFFFFFFFull;
    if (uy == 0 || ux >= 0x7FF0000000000000ull || uy > 0x7FF0000000000000ull) return BKM_NAN;
    if (ux < uy) return x;
    if (ux == uy) return bkm_from_bits(sx);              /* +-0 */
    {
        const double ax = bkm_from_bits(ux), ay = bkm_from_bits(uy), q = ax / ay;
        if (q < 0x1p52) {
            double r = __builtin_fma(-bkm_trunc(q), ay, ax);
            if (r < 0.0) r += ay;
            return bkm_from_bits(bkm_bits(r) | sx);
        }
    }
    ex = (int)(ux >> 52); ey = (int)(uy >> 52);
    mx = ux & 0x000FFFFFFFFFFFFFull; my = uy & 0x000FFFFFFFFFFFFFull;
    if (ex == 0) { sh = __builtin_clzll(mx) - 11; mx <<= sh; ex = 1 - sh; } else mx |= 1ull << 52;
    if (ey == 0) { sh = __builtin_clzll(my) - 11; my <<= sh; ey = 1 - sh; } else my |= 1ull << 52;
    for (; ex > ey; --ex) {
        if (mx >= my) mx -= my;
        mx <<= 1;
    }
    if (mx >= my) mx -= my;
    if (mx == 0) return bkm_from_bits(sx);
    sh = __builtin_clzll(mx) - 11; mx <<= sh; ex -= sh;
    if (ex > 0) ux = (mx & 0x000FFFFFFFFFFFFFull) | ((bkm_u64)ex << 52);
    else ux = mx >> (1 - ex);
    return bkm_from_bits(ux | sx);
}

#endif /* BKM_H */
