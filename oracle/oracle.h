/*
 * oracle.h -- CPU ORACLE for the Blinky warp path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm in
 * engine/NQ/fisheye.c (lensmap build + lensmap apply).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the
 * HIP path.  Nothing under blinky_amd/ may include, link or call it.
 *
 * Parity status: the reference ships no tests/golden vectors for this path
 * ("parity unpinned" by the reference itself).  The oracle is pinned instead
 * against (a) the lensmap hashes recorded from the unmodified reference in
 * SURVEY.md Appendix C and (b) oracle/_ref (the unmodified fisheye.c compiled
 * from /root/reference by oracle/Makefile, see oracle/ref/).
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/engine).
 */
#ifndef BK_ORACLE_H
#define BK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OK_MAX_PLATES 6              /* NQ/fisheye.c:352 */
#define OK_NULL_OFFSET 0xFFFFFFFFu   /* stands for a NULL lens.pixels[] entry */

enum { OK_MAP_NONE = 0, OK_MAP_INVERSE = 1, OK_MAP_FORWARD = 2 };       /* fisheye.c:391 */
enum { OK_ZOOM_NONE = 0, OK_ZOOM_FOV, OK_ZOOM_VFOV, OK_ZOOM_COVER, OK_ZOOM_CONTAIN }; /* :457 */

/* Lens callbacks: return 1 = values, 0 = nil (skip pixel), -1 = malformed (abort).
 * They see the converters below exactly as a Lua script sees the registered C
 * functions (fisheye.c:1494-1537): results are float-rounded. */
typedef int (*ok_inverse_fn)(void *ud, double x, double y, double ray[3]);
typedef int (*ok_forward_fn)(void *ud, double rx, double ry, double rz, double *x, double *y);
/* globe_plate override: return 1 and *plate, or 0 for nil (fisheye.c:1634-1651) */
typedef int (*ok_globe_plate_fn)(void *ud, double rx, double ry, double rz, int *plate);

/* The C functions registered for scripts (fisheye.c:1257-1264), as seen from a
 * lens callback.  Inside the oracle they point at the ok_lua_* restatements;
 * inside oracle/_ref they call the reference's own CtoLUA_* functions. */
typedef struct {
    void (*latlon_to_ray)(void *ctx, double lat, double lon, double out[3]);
    void (*ray_to_latlon)(void *ctx, double x, double y, double z, double *lat, double *lon);
    int  (*plate_to_ray)(void *ctx, double plate, double u, double v, double out[3]);
    void *ctx;
} ok_host;

typedef struct {
    float forward[3], right[3], up[3];   /* vec3_t, fisheye.c:354-356 */
    float fov, dist;                     /* vec_t,  fisheye.c:357-358 */
    uint8_t palette[256];                /* rubix tint LUT, fisheye.c:359 */
    int display;                         /* fisheye.c:360 */
} ok_plate;

typedef struct {
    /* globe (fisheye.c:334-377) */
    ok_plate plates[OK_MAX_PLATES];
    int numplates;
    int platesize;
    ok_globe_plate_fn globe_plate;       /* NULL = argmax of dot products */
    /* lens (fisheye.c:379-451) */
    int map_type;
    double width, height;                /* lens_width / lens_height (0 = absent) */
    double scale;
    int width_px, height_px;
    ok_inverse_fn inverse;
    ok_forward_fn forward;
    void *ud;                            /* passed to the callbacks; = &host for ok_use_lens */
    ok_host host;
    /* zoom (fisheye.c:453-465) */
    int zoom_type, zoom_fov, max_fov, max_vfov;
    /* rubix grid (fisheye.c:467-474) */
    int rubix_numcells;
    double rubix_cell, rubix_pad;
    /* outputs: offset = ptr - globe.pixels, OK_NULL_OFFSET for NULL; tints */
    uint32_t *offsets;
    uint8_t *tints;
} ok_state;

/* pure converters (fisheye.c:1184-1214) */
void ok_latlon_to_ray(double lat, double lon, float ray[3]);
void ok_ray_to_latlon(const float ray[3], double *lat, double *lon);
void ok_plate_uv_to_ray(const ok_state *s, int plate, double u, double v, float ray[3]);

/* the C functions a script sees (fisheye.c:1494-1537): float-rounded doubles */
void ok_lua_latlon_to_ray(double lat, double lon, double out[3]);
void ok_lua_ray_to_latlon(double rx, double ry, double rz, double *lat, double *lon);
int  ok_lua_plate_to_ray(const ok_state *s, double plate, double u, double v, double out[3]);

/* globe loader core (fisheye.c:1796-1869): fwd/up doubles, fov in degrees */
int  ok_set_plate(ok_state *s, int i, const double fwd[3], const double up[3], double fov_deg);

/* zoom (fisheye.c:1293-1386); returns 1 ok / 0 failure, sets s->scale */
int  ok_calc_zoom(ok_state *s);

/* lensmap build (fisheye.c:2367-2397 -> 2084-2124 / 2126-2217), run to completion.
 * Caller provides offsets[W*H], tints[W*H]; they are cleared here the way
 * F_RenderView does (fisheye.c:731-732).  Returns 1 ok, 0 aborted. */
int  ok_create_lensmap(ok_state *s);
/* same, rows [y0,y1) only of the inverse map (used for stripe tests) */
int  ok_build_inverse_rows(ok_state *s, int y0, int y1);

/* lensmap apply (fisheye.c:2406-2424). dst has pitch dst_pitch, origin (x0,y0)
 * = scr_vrect.{x,y}. Unmapped pixels are left untouched. */
void ok_apply(const ok_state *s, const uint8_t *globe, uint8_t *dst, int dst_pitch,
              int x0, int y0, int rubix_on);

/* rubix palette LUTs (fisheye.c:835-908); basepal = 768 bytes */
void ok_create_palmap(ok_state *s, const uint8_t *basepal);

/* f_saveglobe's plate file (WritePCXplate, fisheye.c:1396-1465): PCX header, the plate packed row by row
 * (texels outside the plate's own region become 0xFE unless with_margins), palette.  `plate_pixels` = the
 * plate's ps*ps texels; out must hold ps*ps*2 + 1000 bytes (the reference's Hunk_TempAlloc size).
 * Returns the file length. */
int  ok_write_pcx_plate(const ok_state *s, int plate, int with_margins, const uint8_t *plate_pixels,
                        const uint8_t *basepal, uint8_t *out);

/* helpers for tests */
uint64_t ok_fnv1a64(const void *data, size_t n);
void ok_lcg_fill_plate(uint8_t *dst, size_t n, int plate, int frame);   /* SURVEY 8(d) */

/* hand-transliterated lens/globe scripts (oracle_lenses.c): operation order
 * preserved from game/lua-scripts.  The *_def structs hold the globals a script
 * leaves behind after its chunk ran; callbacks take ud = (ok_host *). */
typedef struct {
    const char *name;
    ok_inverse_fn inverse;               /* NULL if the script defines none */
    ok_forward_fn forward;
    int max_fov, max_vfov;
    double width, height;                /* lens_width / lens_height, 0 = absent */
    const char *onload;
} ok_lens_def;
typedef struct {
    int numplates;
    double forward[OK_MAX_PLATES][3], up[OK_MAX_PLATES][3], fov_deg[OK_MAX_PLATES];
    ok_globe_plate_fn globe_plate;       /* NULL unless the globe script defines globe_plate (globes/fast.lua) */
} ok_globe_def;
/* what a script's chunk can see while it runs: the registered C functions and `numplates` (fisheye.c:1670-1671) */
void ok_set_script_env(const ok_host *host, int numplates);
int  ok_find_lens(const char *name, ok_lens_def *d);
int  ok_find_globe(const char *name, ok_globe_def *g);
/* Returns 1 if known. */
int  ok_use_lens(ok_state *s, const char *name);
void ok_default_host(ok_state *s);       /* host -> ok_lua_* restatements */
int  ok_use_globe(ok_state *s, const char *name);
const char *ok_lens_onload(const char *name);

/* one-call driver used by the tests / CLI: f_globe, f_lens, zoom command, size */
int  ok_configure(ok_state *s, const char *globe, const char *lens, const char *zoomcmd,
                  int W, int H);

#ifdef __cplusplus
}
#endif
#endif
