/*
 * oracle_py.c -- CPU ORACLE (test infrastructure): flat entry points for ctypes.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE                 /* pthread_setaffinity_np, CPU_SET (okpy_time_apply_mt) */
#endif
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* build a lensmap for (globe, lens, zoom) at W x H with the given rubix grid.
 * returns 1 if built; fills offsets/tints (W*H), display[6], scale, numplates, map_type */
int okpy_lensmap(const char *globe, const char *lens, const char *zoomcmd, int W, int H,
                 int numcells, double cell, double pad,
                 uint32_t *offsets, uint8_t *tints, int *display, double *scale,
                 int *numplates, int *map_type)
{
    ok_state s;
    int i, ok;
    if (!ok_configure(&s, globe, lens, zoomcmd, W, H)) return -1;
    s.rubix_numcells = numcells; s.rubix_cell = cell; s.rubix_pad = pad;
    s.offsets = offsets; s.tints = tints;
    ok = ok_create_lensmap(&s);
    for (i = 0; i < OK_MAX_PLATES; ++i) display[i] = i < s.numplates ? s.plates[i].display : 0;
    *scale = s.scale; *numplates = s.numplates; *map_type = s.map_type;
    return ok;
}

/* plates of a transliterated globe, in LUA_load_globe's float form (13 floats per plate:
 * forward[3] right[3] up[3] fov dist) */
int okpy_globe(const char *globe, float *out13, int *numplates)
{
    ok_state s;
    int i;
    memset(&s, 0, sizeof s);
    if (!ok_use_globe(&s, globe)) return 0;
    for (i = 0; i < s.numplates; ++i) {
        float *o = out13 + 13 * i;
        memcpy(o, s.plates[i].forward, 12); memcpy(o + 3, s.plates[i].right, 12);
        memcpy(o + 6, s.plates[i].up, 12); o[9] = s.plates[i].fov; o[10] = s.plates[i].dist;
        o[11] = o[12] = 0;
    }
    *numplates = s.numplates;
    return 1;
}

/* render_lensmap over a table of `rows` rows */
void okpy_apply(const uint32_t *offsets, const uint8_t *tints, int W, int rows,
                const uint8_t *globe, uint8_t *dst, int dst_pitch, int x0, int y0,
                int rubix_on, const uint8_t *pal /* [6][256] or NULL */)
{
    ok_state s;
    int i;
    memset(&s, 0, sizeof s);
    s.width_px = W; s.height_px = rows;
    s.offsets = (uint32_t *)offsets; s.tints = (uint8_t *)tints;
    if (pal) for (i = 0; i < OK_MAX_PLATES; ++i) memcpy(s.plates[i].palette, pal + 256 * i, 256);
    ok_apply(&s, globe, dst, dst_pitch, x0, y0, rubix_on);
}

/* the same render_lensmap restated row-parallel over `nthreads` host threads (one band of rows
 * per thread, threads pinned round-robin to the CPUs this process may use) - the "all host cores" CPU figure
 * next to the faithful single-threaded one.  Returns the best wall time in seconds over `reps` calls. */
#include <pthread.h>
#include <sched.h>
#include <time.h>
typedef struct {
    ok_state s;
    const uint8_t *globe;
    uint8_t *dst;
    int dst_pitch, r0, reps, index, cpu;
    pthread_barrier_t *bar;
    double *best;
} okpy_band;
static void *okpy_band_main(void *arg)
{
    okpy_band *b = (okpy_band *)arg;
    int r;
    if (b->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(b->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    for (r = 0; r < b->reps; ++r) {
        struct timespec t0, t1;
        pthread_barrier_wait(b->bar);
        if (b->index == 0) clock_gettime(CLOCK_MONOTONIC, &t0);
        ok_apply(&b->s, b->globe, b->dst, b->dst_pitch, 0, b->r0, 0);
        pthread_barrier_wait(b->bar);
        if (b->index == 0) {
            double dt;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
            if (dt < *b->best) *b->best = dt;
        }
    }
    return NULL;
}
double okpy_time_apply_mt(const uint32_t *offsets, const uint8_t *tints, int W, int rows, const uint8_t *globe,
                          uint8_t *dst, int dst_pitch, int reps, int nthreads)
{
    double best = 1e30;
    pthread_barrier_t bar;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    okpy_band *bands = (okpy_band *)calloc((size_t)nthreads, sizeof *bands);
    cpu_set_t allowed;
    int cpus[1024], ncpu = 0, c, i;
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0)
        for (c = 0; c < CPU_SETSIZE && ncpu < 1024; ++c) if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
    for (i = 0; i < nthreads; ++i) {
        okpy_band *b = &bands[i];
        const int r0 = (int)((long long)rows * i / nthreads), r1 = (int)((long long)rows * (i + 1) / nthreads);
        memset(&b->s, 0, sizeof b->s);
        b->s.width_px = W; b->s.height_px = r1 - r0;
        b->s.offsets = (uint32_t *)offsets + (size_t)r0 * W; b->s.tints = (uint8_t *)tints + (size_t)r0 * W;
        b->globe = globe; b->dst = dst; b->dst_pitch = dst_pitch; b->r0 = r0; b->reps = reps; b->index = i;
        b->cpu = ncpu ? cpus[i % ncpu] : -1;
        b->bar = &bar; b->best = &best;
        pthread_create(&th[i], NULL, okpy_band_main, b);
    }
    for (i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    pthread_barrier_destroy(&bar);
    free(th); free(bands);
    return best;
}

void okpy_palmap(const uint8_t *basepal, uint8_t *out /* [6][256] */)
{
    ok_state s;
    int i;
    memset(&s, 0, sizeof s);
    ok_create_palmap(&s, basepal);
    for (i = 0; i < OK_MAX_PLATES; ++i) memcpy(out + 256 * i, s.plates[i].palette, 256);
}

/* f_saveglobe: the PCX file of one plate of a named globe at platesize ps */
int okpy_pcx_plate(const char *globe, int ps, int plate, int with_margins, const uint8_t *plate_pixels,
                   const uint8_t *basepal, uint8_t *out)
{
    ok_state s;
    memset(&s, 0, sizeof s);
    if (!ok_use_globe(&s, globe)) return -1;
    s.platesize = ps;
    return ok_write_pcx_plate(&s, plate, with_margins, plate_pixels, basepal, out);
}

/* evaluate a hand-transliterated callback: which 0 = lens_inverse(x,y), 1 = lens_forward(x,y,z).
 * returns 1 values written, 0 nil, -1 unknown lens / missing callback */
int okpy_eval(const char *lens, int which, double x, double y, double z, double *out)
{
    ok_state s;
    ok_lens_def d;
    memset(&s, 0, sizeof s);
    ok_default_host(&s);
    ok_set_script_env(&s.host, 6);
    if (!ok_find_lens(lens, &d)) return -1;
    if (which == 0) { if (!d.inverse) return -1; return d.inverse(&s.host, x, y, out); }
    if (!d.forward) return -1;
    return d.forward(&s.host, x, y, z, &out[0], &out[1]);
}

int okpy_lens_def(const char *lens, int *has_inverse, int *has_forward, int *max_fov, int *max_vfov,
                  double *width, double *height, char *onload, int cap)
{
    ok_lens_def d;
    ok_state s;
    memset(&s, 0, sizeof s);
    ok_default_host(&s);
    ok_set_script_env(&s.host, 6);
    if (!ok_find_lens(lens, &d)) return 0;
    *has_inverse = d.inverse != NULL; *has_forward = d.forward != NULL;
    *max_fov = d.max_fov; *max_vfov = d.max_vfov; *width = d.width; *height = d.height;
    strncpy(onload, d.onload ? d.onload : "", (size_t)cap - 1);
    onload[cap - 1] = 0;
    return 1;
}

/* Generic-lens variant: the lens callbacks are supplied by the caller (the tests pass ctypes
 * callbacks that evaluate the real Lua script with an interpreter on the platform libm), the rest -
 * globe, zoom, build, grid - is the fisheye.c restatement above.  Lens globals come as arguments. */
typedef int (*okpy_inv_cb)(double x, double y, double *out3);                 /* 1 values, 0 nil, -1 error */
typedef int (*okpy_fwd_cb)(double x, double y, double z, double *out2);
static okpy_inv_cb g_inv_cb;
static okpy_fwd_cb g_fwd_cb;
static int cb_inverse(void *ud, double x, double y, double ray[3]) { (void)ud; return g_inv_cb(x, y, ray); }
static int cb_forward(void *ud, double x, double y, double z, double *ox, double *oy)
{
    double o[2];
    int rc = g_fwd_cb(x, y, z, o);
    (void)ud;
    *ox = o[0]; *oy = o[1];
    return rc;
}

int okpy_lensmap_cb(const char *globe, okpy_inv_cb inv, okpy_fwd_cb fwd, int map_type, int max_fov, int max_vfov,
                    double lens_width, double lens_height, const char *zoomcmd, int W, int H,
                    uint32_t *offsets, uint8_t *tints, int *display, double *scale, int *numplates)
{
    ok_state s;
    int i, ok;
    memset(&s, 0, sizeof s);
    ok_default_host(&s);
    s.rubix_numcells = 10; s.rubix_cell = 4; s.rubix_pad = 1;
    if (!ok_use_globe(&s, globe)) return -1;
    g_inv_cb = inv; g_fwd_cb = fwd;
    s.inverse = inv ? cb_inverse : NULL;
    s.forward = fwd ? cb_forward : NULL;
    s.ud = &s.host;
    s.map_type = map_type; s.max_fov = max_fov; s.max_vfov = max_vfov;
    s.width = lens_width; s.height = lens_height;
    s.zoom_type = OK_ZOOM_NONE; s.zoom_fov = 0;
    if (zoomcmd && !strncmp(zoomcmd, "f_fov ", 6)) { s.zoom_type = OK_ZOOM_FOV; s.zoom_fov = (int)atof(zoomcmd + 6); }
    else if (zoomcmd && !strncmp(zoomcmd, "f_vfov ", 7)) { s.zoom_type = OK_ZOOM_VFOV; s.zoom_fov = (int)atof(zoomcmd + 7); }
    else if (zoomcmd && !strcmp(zoomcmd, "f_cover")) s.zoom_type = OK_ZOOM_COVER;
    else if (zoomcmd && !strcmp(zoomcmd, "f_contain")) s.zoom_type = OK_ZOOM_CONTAIN;
    s.width_px = W; s.height_px = H;
    s.platesize = W < H ? W : H;
    s.offsets = offsets; s.tints = tints;
    ok = ok_create_lensmap(&s);
    for (i = 0; i < OK_MAX_PLATES; ++i) display[i] = i < s.numplates ? s.plates[i].display : 0;
    *scale = s.scale; *numplates = s.numplates;
    return ok;
}
