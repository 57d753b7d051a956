/*
 * oracle_lenses.c -- CPU ORACLE (test infrastructure only).
 *
 * Hand transliterations of a few lens / globe scripts from
 * /root/reference/game/lua-scripts into C, preserving Lua 5.2's evaluation
 * order (all arithmetic in double, `^` = pow(), math.xxx = libm xxx).  They are
 * deliberately independent of the product's Lua front-end so that a front-end
 * bug cannot cancel out in a parity test.  Same method as the survey probe
 * (SURVEY.md Appendix C), whose reference-derived hashes pin these.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LUA_PI 3.14159265358979323846   /* math.pi */

/* calls through volatile pointers so gcc cannot constant-fold / strength-reduce
 * what a real Lua VM would evaluate with a libm call at run time */
#ifdef OK_PORTABLE_LIBM   /* liboracle_bkm.so, see oracle.c */
#include "../blinky_amd/csrc/bkm.h"
#define pow bkm_pow
#define sqrt bkm_sqrt
#define sin bkm_sin
#define cos bkm_cos
#define tan bkm_tan
#define asin bkm_asin
#define acos bkm_acos
#define atan bkm_atan
#define atan2 bkm_atan2
#define sinh bkm_sinh
#define cosh bkm_cosh
#define tanh bkm_tanh
#define exp bkm_exp
#endif
static double (*volatile lua_pow)(double, double) = pow;
static double (*volatile lua_sqrt)(double) = sqrt;
/* A Lua VM makes one libm call per math.xxx; gcc -O2 would fuse sin(x)/cos(x) pairs of the
 * transliterations into sincos(), whose glibc results differ from sin()/cos() in the last bit
 * for ~0.1 % of arguments.  Route every call through a volatile pointer to keep them separate.
 * (oracle.c, which restates the reference's own C code, is deliberately left to gcc: there
 * the reference binary gets the same fusion - oracle/_ref arbitrates.) */
static double (*volatile lua_sin)(double) = sin;
static double (*volatile lua_cos)(double) = cos;
static double (*volatile lua_tan)(double) = tan;
static double (*volatile lua_asin)(double) = asin;
static double (*volatile lua_acos)(double) = acos;
static double (*volatile lua_atan)(double) = atan;
static double (*volatile lua_atan2)(double, double) = atan2;
static double (*volatile lua_sinh)(double) = sinh;
static double (*volatile lua_cosh)(double) = cosh;
static double (*volatile lua_tanh)(double) = tanh;
static double (*volatile lua_exp)(double) = exp;
#undef sin
#undef cos
#undef tan
#undef asin
#undef acos
#undef atan
#undef atan2
#undef sinh
#undef cosh
#undef tanh
#undef exp
#undef sqrt
#undef pow
#define pow lua_pow
#define sin lua_sin
#define cos lua_cos
#define tan lua_tan
#define asin lua_asin
#define acos lua_acos
#define atan lua_atan
#define atan2 lua_atan2
#define sinh lua_sinh
#define cosh lua_cosh
#define tanh lua_tanh
#define exp lua_exp
#define sqrt lua_sqrt

/* The three C functions a script may call (fisheye.c:1257-1264) are reached
 * through ok_host so that the same transliterations run both inside the oracle
 * (-> ok_lua_* restatements) and inside oracle/_ref (-> the reference's own
 * CtoLUA_* functions, through the fake Lua stack). */
#define H_LATLON_TO_RAY(ud, lat, lon, out) ((const ok_host *)(ud))->latlon_to_ray(((const ok_host *)(ud))->ctx, lat, lon, out)
#define H_RAY_TO_LATLON(ud, x, y, z, lat, lon) ((const ok_host *)(ud))->ray_to_latlon(((const ok_host *)(ud))->ctx, x, y, z, lat, lon)

/* ---- panini.lua:1-25 --------------------------------------------------------- */
static const double panini_d = 1;

static int panini_inverse(void *ud, double x, double y, double ray[3])
{
    double d = panini_d;
    double k = x * x / ((d + 1) * (d + 1));                 /* panini.lua:9 */
    double dscr = k * k * d * d - (k + 1) * (k * d * d - 1);  /* :10 */
    double clon = (-k * d + sqrt(dscr)) / (k + 1);          /* :11 */
    double S = (d + 1) / (d + clon);                        /* :12 */
    double lon = atan2(x, S * clon);                        /* :13 */
    double lat = atan2(y, S);                               /* :14 */
    H_LATLON_TO_RAY(ud, lat, lon, ray);                    /* :16 */
    return 1;
}

static int panini_forward(void *ud, double x, double y, double z, double *ox, double *oy)
{
    double lat, lon, S, d = panini_d;
    H_RAY_TO_LATLON(ud, x, y, z, &lat, &lon);              /* :20 */
    S = (d + 1) / (d + cos(lon));                           /* :21 */
    *ox = S * sin(lon);                                     /* :22 */
    *oy = S * tan(lat);                                     /* :23 */
    return 1;
}

/* ---- stereographic.lua:1-23 -------------------------------------------------- */
static const double stereo_angleScale = 0.5;

static int stereographic_inverse(void *ud, double x, double y, double ray[3])
{
    double r = sqrt(x * x + y * y);                         /* :9 */
    double theta = atan(r) / stereo_angleScale;             /* :10 */
    double s = sin(theta);                                  /* :12 */
    ray[0] = x / r * s;                                     /* :13 */
    ray[1] = y / r * s;
    ray[2] = cos(theta);
    return 1;
}

static int stereographic_forward(void *ud, double x, double y, double z, double *ox, double *oy)
{
    double theta = acos(z);                                 /* :17 */
    double r = tan(theta * stereo_angleScale);              /* :19 */
    double c = r / sqrt(x * x + y * y);                     /* :21 */
    *ox = x * c;
    *oy = y * c;
    return 1;
}

/* ---- hammer.lua:1-24 --------------------------------------------------------- */
static int hammer_inverse(void *ud, double x, double y, double ray[3])
{
    double z, lon, lat;
    if (x * x / 8 + y * y / 2 > 1)                          /* :10 */
        return 0;
    z = sqrt(1 - 0.0625 * x * x - 0.25 * y * y);            /* :13 */
    lon = 2 * atan(z * x / (2 * (2 * z * z - 1)));          /* :14 */
    lat = asin(z * y);                                      /* :15 */
    H_LATLON_TO_RAY(ud, lat, lon, ray);
    return 1;
}

static int hammer_forward(void *ud, double x, double y, double z, double *ox, double *oy)
{
    double lat, lon;
    H_RAY_TO_LATLON(ud, x, y, z, &lat, &lon);
    *ox = 2 * lua_sqrt(2) * cos(lat) * sin(lon * 0.5) / sqrt(1 + cos(lat) * cos(lon * 0.5)); /* :21 */
    *oy = lua_sqrt(2) * sin(lat) / sqrt(1 + cos(lat) * cos(lon * 0.5));                     /* :22 */
    return 1;
}

/* ---- eckert5.lua:1-15 (forward only) ----------------------------------------- */
static int eckert5_forward(void *ud, double x, double y, double z, double *ox, double *oy)
{
    double lat, lon;
    H_RAY_TO_LATLON(ud, x, y, z, &lat, &lon);
    *ox = lon * (1 + cos(lat)) / 2;                         /* :12 */
    *oy = lat;                                              /* :13 */
    return 1;
}

/* ---- quincuncial.lua:1-206 --------------------------------------------------- */
static const double q_eps = 0.0001;                         /* :1 */
#define q_halfpi (LUA_PI / 2)                               /* :2 */
static double q_sqrt2, q_sqrt22;                            /* :70-71 */
static const double q_m = 1.0 / 2;                          /* :72 */
static const double q_ke = 1.85407467730137;                /* :73 */

static double q_asqrt(double x) { return x > 0 ? sqrt(x) : 0; }   /* :9-14 */

/* :15-63, returns sn, cn, dn (the 4th value phi is never used by callers) */
static void q_ellipj(double u, double m, double *sn, double *cn, double *dn)
{
    double ai, b, phi, t, twon;
    double a[10], c[10];
    int i, k;
    if (m < q_eps) {                                        /* :17-25 */
        t = sin(u);
        b = cos(u);
        ai = .25 * m * (u - t * b);
        *sn = t - ai * b;
        *cn = b + ai * t;
        *dn = 1 - .5 * m * t * t;
        return;
    }
    if (m >= 1 - q_eps) {                                   /* :26-36 */
        ai = .25 * (1 - m);
        b = cosh(u);
        t = tanh(u);
        phi = 1 / b;
        twon = b * sinh(u);
        *sn = t + ai * (twon - u) / (b * b);
        *cn = phi - ai * t * phi * (twon - u);
        *dn = phi + ai * t * phi * (twon + u);
        return;
    }
    for (k = 1; k <= 9; ++k) { a[k] = 0; c[k] = 0; }        /* :38-39 */
    a[1] = 1;
    c[1] = sqrt(m);
    i = 1;
    b = sqrt(1 - m);                                        /* :41 */
    twon = 1;
    while (fabs(c[i] / a[i]) > q_eps && i < 9) {            /* :44 */
        ai = a[i];
        i = i + 1;
        c[i] = .5 * (ai - b);
        a[i] = .5 * (ai + b);
        b = q_asqrt(ai * b);
        twon = twon * 2;
    }
    phi = twon * a[i] * u;                                  /* :53 */
    do {                                                    /* :54-59 */
        b = phi;
        t = c[i] * sin(b) / a[i];
        phi = .5 * (asin(t) + phi);
        i = i - 1;
    } while (!(i == 1));
    t = cos(phi);                                           /* :61 */
    *sn = sin(phi);
    *cn = t;
    *dn = t / cos(phi - b);
}

/* :75-104 */
static void q_cnrectify(double x, double y, double *latp, double *longd)
{
    double xpr = q_ke * (q_sqrt22 * x - q_sqrt22 * y) / q_sqrt2 + q_ke;   /* :79 */
    double ypr = q_ke * (q_sqrt22 * x + q_sqrt22 * y) / q_sqrt2;          /* :80 */
    double x1, y1;
    if (fabs(ypr) < q_eps) {                                /* :87 */
        double sni, cni, dni;
        q_ellipj(xpr, q_m, &sni, &cni, &dni);
        x1 = cni;
        y1 = 0.0;
    } else {
        double s, c, d, s1, c1, d1, delta;
        q_ellipj(xpr, q_m, &s, &c, &d);                     /* :94 */
        q_ellipj(ypr, 1 - q_m, &s1, &c1, &d1);              /* :95 */
        delta = lua_pow(c1, 2) + q_m * lua_pow(s, 2) * lua_pow(s1, 2);   /* :96 */
        x1 = (c * c1) / delta;                              /* :97 */
        y1 = -(s * d * s1 * d1) / delta;                    /* :98 */
    }
    *longd = atan2(y1, x1);                                 /* :101 */
    *latp = 2 * atan2(sqrt(x1 * x1 + y1 * y1), 1) - q_halfpi;   /* :102 */
}

static void q_rotate(double a, double b, double angle, double *a0, double *b0)   /* :150-156 */
{
    double c = cos(angle);
    double s = sin(angle);
    *a0 = a * c - b * s;
    *b0 = a * s + b * c;
}

static int q_inverse_intermediate(void *ud, double x, double y, double ray[3])              /* :158-169 */
{
    double lat, lon, r[3];
    if (fabs(x) > 2 || fabs(y) > 1)
        return 0;
    x = x + 1;
    q_cnrectify(x, y, &lat, &lon);
    H_LATLON_TO_RAY(ud, lat, -lon, r);                     /* :164 */
    ray[0] = r[0];                                          /* :167  x1,y1,z1 = x0, z0, -y0 */
    ray[1] = r[2];
    ray[2] = -r[1];
    return 1;
}

static int quincuncial_inverse(void *ud, double x, double y, double ray[3])       /* :171-206 */
{
    double x0, y0;
        if (fabs(x) > q_sqrt2 || fabs(y) > q_sqrt2)
        return 0;
    if (fabs(x) + fabs(y) < q_sqrt2) {                      /* front */
        q_rotate(x, y, LUA_PI / 4, &x0, &y0);
        x0 = x0 - 1;
    } else if (x > 0 && y < 0) {                            /* lower right */
        q_rotate(x, y, LUA_PI / 4, &x0, &y0);
        x0 = x0 - 1;
    } else if (x < 0 && y > 0) {                            /* upper left */
        q_rotate(x, y, LUA_PI / 4, &x0, &y0);
        x0 = x0 + 3;
    } else if (x < 0 && y < 0) {                            /* lower left */
        q_rotate(x, y, LUA_PI / 4 + LUA_PI, &x0, &y0);
        x0 = x0 + 1; y0 = y0 - 2;
    } else {                                                /* upper right */
        q_rotate(x, y, LUA_PI / 4 + LUA_PI, &x0, &y0);
        x0 = x0 + 1; y0 = y0 + 2;
    }
    return q_inverse_intermediate(ud, x0, y0, ray);
}

/* ---- registry ---------------------------------------------------------------- */

static void lens_globals(const char *name, ok_lens_def *d)
{
    memset(d, 0, sizeof(*d));
    d->name = name;
    if (!strcmp(name, "panini")) {
        d->inverse = panini_inverse; d->forward = panini_forward;
        d->max_fov = 360; d->max_vfov = 180; d->onload = "f_fov 180";
    } else if (!strcmp(name, "stereographic")) {
        d->inverse = stereographic_inverse; d->forward = stereographic_forward;
        d->max_fov = 360; d->max_vfov = 360; d->onload = "f_fov 180";
    } else if (!strcmp(name, "hammer")) {
        d->inverse = hammer_inverse; d->forward = hammer_forward;
        d->max_fov = 360; d->max_vfov = 180; d->onload = "f_contain";
        d->width = 2 * lua_sqrt(2) * 2;                     /* hammer.lua:4 */
        d->height = lua_sqrt(2) * 2;                        /* hammer.lua:5 */
    } else if (!strcmp(name, "quincuncial")) {
        q_sqrt2 = lua_sqrt(2);
        q_sqrt22 = q_sqrt2 / 2;
        d->inverse = quincuncial_inverse; d->onload = "f_contain";
        d->height = 2 * q_sqrt2;                            /* quincuncial.lua:106-107 */
        d->width = 2 * q_sqrt2;
    } else if (!strcmp(name, "eckert5")) {
        d->forward = eckert5_forward; d->onload = "f_contain";
        d->max_fov = 360; d->max_vfov = 180;
        d->width = LUA_PI * 2;                              /* eckert5.lua:5-6 */
        d->height = LUA_PI;
    } else {
        d->name = NULL;
    }
}

/* runs the "chunk": returns the globals the script would leave behind */
int ok_find_lens(const char *name, ok_lens_def *d)
{
    lens_globals(name, d);
    return d->name != NULL;
}

int ok_find_globe(const char *name, ok_globe_def *g)
{
    memset(g, 0, sizeof(*g));
    if (!strcmp(name, "cube")) {                            /* globes/cube.lua:3-10 */
        static const double f[6][3] = {{0,0,1},{1,0,0},{-1,0,0},{0,0,-1},{0,1,0},{0,-1,0}};
        static const double u[6][3] = {{0,1,0},{0,1,0},{0,1,0},{0,1,0},{0,0,-1},{0,0,1}};
        int i;
        for (i = 0; i < 6; ++i) {
            memcpy(g->forward[i], f[i], sizeof f[i]);
            memcpy(g->up[i], u[i], sizeof u[i]);
            g->fov_deg[i] = 90;
        }
        g->numplates = 6;
        return 1;
    }
    if (!strcmp(name, "trism")) {                           /* globes/trism.lua:2-8 */
        double c = cos(LUA_PI / 6), sn = sin(LUA_PI / 6);
        double f[5][3] = {{-c,0,sn},{c,0,sn},{0,0,-1},{0,1,0},{0,-1,0}};
        double u[5][3] = {{0,1,0},{0,1,0},{0,1,0},{0,0,-1},{0,0,-1}};
        double fov[5] = {120,120,120,128,128};
        int i;
        for (i = 0; i < 5; ++i) {
            memcpy(g->forward[i], f[i], sizeof f[i]);
            memcpy(g->up[i], u[i], sizeof u[i]);
            g->fov_deg[i] = fov[i];
        }
        g->numplates = 5;
        return 1;
    }
    return 0;
}

/* LUA_load_lens (fisheye.c:1659-1750) on the transliterated globals */
int ok_use_lens(ok_state *s, const char *name)
{
    ok_lens_def d;
    if (!ok_find_lens(name, &d)) return 0;
    s->inverse = d.inverse; s->forward = d.forward; s->ud = &s->host;
    s->width = d.width; s->height = d.height;               /* :1741-1747 */
    s->max_fov = d.max_fov; s->max_vfov = d.max_vfov;       /* :1733-1739 */
    s->map_type = OK_MAP_NONE;
    if (s->inverse) s->map_type = OK_MAP_INVERSE;           /* :1688-1709 inverse preferred */
    else if (s->forward) s->map_type = OK_MAP_FORWARD;
    return 1;
}

const char *ok_lens_onload(const char *name)
{
    ok_lens_def d;
    return ok_find_lens(name, &d) && d.onload ? d.onload : "";
}

/* LUA_load_globe (fisheye.c:1752-1875) on the transliterated plates table */
int ok_use_globe(ok_state *s, const char *name)
{
    ok_globe_def g;
    int i;
    s->globe_plate = NULL;
    s->numplates = 0;
    if (!ok_find_globe(name, &g)) return 0;
    for (i = 0; i < g.numplates; ++i)
        if (!ok_set_plate(s, i, g.forward[i], g.up[i], g.fov_deg[i])) return 0;
    s->numplates = g.numplates;
    return 1;
}

/* "f_globe G; f_lens L; <zoom>" followed by one F_RenderView at W x H, minus the
 * build itself (fisheye.c:1138-1161, 1061-1103, 1032-1058, 955-965, 704-707) */
int ok_configure(ok_state *s, const char *globe, const char *lens, const char *zoomcmd,
                 int W, int H)
{
    memset(s, 0, sizeof(*s));
    ok_default_host(s);
    s->rubix_numcells = 10; s->rubix_cell = 4; s->rubix_pad = 1;       /* fisheye.c:672 */
    if (!ok_use_globe(s, globe)) return 0;
    if (!ok_use_lens(s, lens)) return 0;
    if (!zoomcmd || !*zoomcmd) zoomcmd = ok_lens_onload(lens);
    s->zoom_type = OK_ZOOM_NONE; s->zoom_fov = 0;
    if (!strncmp(zoomcmd, "f_fov ", 6)) { s->zoom_type = OK_ZOOM_FOV; s->zoom_fov = (int)atof(zoomcmd + 6); }
    else if (!strncmp(zoomcmd, "f_vfov ", 7)) { s->zoom_type = OK_ZOOM_VFOV; s->zoom_fov = (int)atof(zoomcmd + 7); }
    else if (!strcmp(zoomcmd, "f_cover")) s->zoom_type = OK_ZOOM_COVER;
    else if (!strcmp(zoomcmd, "f_contain")) s->zoom_type = OK_ZOOM_CONTAIN;
    s->width_px = W; s->height_px = H;
    s->platesize = W < H ? W : H;                                       /* fisheye.c:707 */
    return 1;
}
