/* bk_hostmod_bkm.h -- stands in for bkm.h when the generated lens code is compiled for the HOST (the re-derivation of the
 * entries a build flagged, bk_lens.cpp: host module): every bkm_* name is the PLATFORM libm's function of that name - what
 * the reference's Lua VM calls on this machine (lmathlib.c: math.sin == sin ...), and what the script interpreter's
 * math_platform() table holds.  The build compiles this unit with -fno-builtin, so that no call is folded at compile time
 * (a folded sin(0.5) is the correctly rounded value; glibc's may be its neighbour). */
#ifndef BKM_H
#define BKM_H
#include <math.h>
#define BKM_INF (__builtin_inf())
#define BKM_NAN (__builtin_nan(""))
static inline int bkm_isinf(double x) { return __builtin_fabs(x) == BKM_INF; }
static inline int bkm_isnan(double x) { return x != x; }
static inline double bkm_fabs(double x) { return __builtin_fabs(x); }
static inline double bkm_copysign(double x, double s) { return __builtin_copysign(x, s); }
static inline double bkm_trunc(double x) { return __builtin_trunc(x); }
static inline double bkm_floor(double x) { return __builtin_floor(x); }
static inline double bkm_ceil(double x) { return __builtin_ceil(x); }
static inline double bkm_rint(double x) { return __builtin_rint(x); }
static inline double bkm_sqrt(double x) { return sqrt(x); }
static inline double bkm_fmod(double x, double y) { return fmod(x, y); }
static inline double bkm_sin(double x) { return sin(x); }
static inline double bkm_cos(double x) { return cos(x); }
static inline void bkm_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }   /* two calls, as the reference makes them */
static inline double bkm_tan(double x) { return tan(x); }
static inline double bkm_asin(double x) { return asin(x); }
static inline double bkm_acos(double x) { return acos(x); }
static inline double bkm_atan(double x) { return atan(x); }
static inline double bkm_atan2(double y, double x) { return atan2(y, x); }
static inline double bkm_sinh(double x) { return sinh(x); }
static inline double bkm_cosh(double x) { return cosh(x); }
static inline double bkm_tanh(double x) { return tanh(x); }
static inline double bkm_exp(double x) { return exp(x); }
static inline double bkm_log(double x) { return log(x); }
static inline double bkm_log10(double x) { return log10(x); }
static inline double bkm_pow(double x, double y) { return pow(x, y); }
#endif
