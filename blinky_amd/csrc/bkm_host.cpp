// bkm_host.cpp -- host build of the portable libm (bkm.h) as a tiny shared library, used by
// tests/test_bkm.py to measure it against mpmath and by the GPU test that checks that the
// device build returns bit-identical results.
#include "bkm.h"
#define W1(n) extern "C" double bkmh_##n(double x) { return bkm_##n(x); }
#define W2(n) extern "C" double bkmh_##n(double x, double y) { return bkm_##n(x, y); }
W1(sin) W1(cos) W1(tan) W1(asin) W1(acos) W1(atan) W1(sinh) W1(cosh) W1(tanh) W1(exp) W1(log) W1(log10)
W2(atan2) W2(pow) W2(fmod)
extern "C" void bkmh_map1(const char *name, const double *x, double *out, long n)
{
    double (*f)(double) = nullptr;
#define SEL(nm) if (!__builtin_strcmp(name, #nm)) f = bkmh_##nm;
    SEL(sin) SEL(cos) SEL(tan) SEL(asin) SEL(acos) SEL(atan) SEL(sinh) SEL(cosh) SEL(tanh) SEL(exp) SEL(log) SEL(log10)
    if (!f) return;
    for (long i = 0; i < n; ++i) out[i] = f(x[i]);
}
extern "C" void bkmh_map2(const char *name, const double *x, const double *y, double *out, long n)
{
    double (*f)(double, double) = nullptr;
    if (!__builtin_strcmp(name, "atan2")) f = bkmh_atan2;
    if (!__builtin_strcmp(name, "pow")) f = bkmh_pow;
    if (!__builtin_strcmp(name, "fmod")) f = bkmh_fmod;
    if (!f) return;
    for (long i = 0; i < n; ++i) out[i] = f(x[i], y[i]);
}
