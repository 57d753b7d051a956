/* bk_build_params.h -- kernel-argument block of the lensmap build kernels.  Included by the
 * host (bk_lens.cpp) and embedded verbatim into the hiprtc translation unit, so both sides
 * see one definition.  Plain C types only. */
#ifndef BK_BUILD_PARAMS_H
#define BK_BUILD_PARAMS_H

typedef struct {
    float forward[3], right[3], up[3];   /* vec3_t, as LUA_load_globe leaves them (fisheye.c:1818-1850) */
    float fov, dist;                     /* fisheye.c:1858, 1868 */
    double dist64;                       /* 0.5 / tan(fov/2) recomputed in double (fisheye.c:2060) */
} BkPlateDev;

typedef struct {
    int W, H;                /* lens.width_px / height_px */
    int row0, rows;          /* owned output rows [row0, row0+rows) */
    int ps, gp, ph;          /* platesize; padded plate width / height of the device globe (bk_texel_offset) */
    int numplates, has_globe_plate;
    double scale;            /* lens.scale */
    double rubix_block, rubix_pad, rubix_unit_px;   /* set_lensmap_grid constants (fisheye.c:1938-1948) */
    /* ... and what they say about every texel column / row p of a plate: bit p = "p lies on a grid line", i.e.
     * fmod((double)p / rubix_unit_px, rubix_block) < rubix_pad (fisheye.c:1950-1957), evaluated once per build on the host - exact
     * operations, the same on any machine - instead of two divisions and two fmods per pixel in the kernel.  grid_n = ps when the
     * table is filled (ps <= 8192), 0 = evaluate the formula. */
    unsigned int grid_bits[256];
    int grid_n;
    BkPlateDev plates[6];
    unsigned int *offsets;   /* [rows][W] device-layout offsets (bk_texel_offset), 0xFFFFFFFF = NULL */
    unsigned char *tints;    /* [rows][W] */
    int *display;            /* [6] */
    int *err;                /* [1] OR of BK_ERR_* bits */
    /* forward build (fisheye.c:2126-2338) */
    unsigned int *fwd_key_px;    /* [rows][W] 1 + order index of the last writer (0 = none), max-reduced */
    unsigned int *fwd_key_tint;  /* [rows][W] same, over writers that were off the rubix grid */
    int *corner_xy;          /* [numplates][ps+1][ps+1][2] screen coords of the texel corners */
    unsigned char *corner_ok;/* [numplates][ps+1][ps+1] */
    /* pixels / corners / texels whose discrete outcome depends on libm's last bits (bk_device_rt.h): their
     * indices are appended here and re-evaluated by the host interpreter on the platform libm */
    unsigned int *flag_list; /* [flag_cap][4]: id, then what the device derived (offset, tint, 0 | sx, sy, ok | own, 0, 0) */
    unsigned int *flag_count;/* [1] total flagged (may exceed flag_cap: the host then retries with a larger list) */
    unsigned int flag_cap;
    /* forward build, second pass only: host-decided answers to "does this texel's ray select its own plate" for the
     * texels the first pass flagged (globe_plate scripts): (texel id << 1 | answer), ascending */
    const unsigned int *ovr_list;
    unsigned int ovr_count;
    /* forward build: what the kernels would otherwise divide out per texel corner / per texel / per quad edge, evaluated once on the
     * host with the same IEEE operations (device memory; the host module of the flagged entries computes instead and never reads them):
     * fwd_quot[a * 21 + d] = (double)a / (double)d for 0 <= a <= 20, 1 <= d <= 20 - draw_quad's edge interpolation (fisheye.c:2313);
     * fwd_uv[i] = (float)(((double)i - 0.5) / ps - 0.5), i in 0..ps - a texel corner's offset along right (negated: along up) in
     * plate_uv_to_ray (fisheye.c:1205-1211, 2229); fwd_uv[ps + 1 + i] = (float)((double)i / ps - 0.5) - a texel's own ray (:2193). */
    const double *fwd_quot;
    const float *fwd_uv;
    /* forward build: tile_own[(plate * nt + ty) * nt + tx], nt = ceil(ps / 16) - 1 when EVERY texel of that 16 x 16 tile selects its own
     * plate with room to spare (bk_forward_tiles), so that the quad pass need not ask texel by texel; null: ask */
    const unsigned char *tile_own;
    int *counters_out;       /* forward build, resolve pass: where to leave a copy of the 2 x 9 counters at display[] (pinned host memory), or null */
    double inv_scale_up;     /* >= 1 / scale: turns an error bound in lens units into screen pixels without a division */
    /* inverse build: 1 + scan key of the FIRST pixel (in the reference's scan order: rows bottom-up, pixels left to right,
     * fisheye.c:2093-2103) whose callback returned a malformed result - key = ly * W + (W - 1 - lx), max-reduced; 0 = none */
    unsigned int *first_bad;
} BkBuildParams;

/* Device globe layout.  A plate is gp = round_up(ps,64) texels wide and ph = round_up(ps,8) high and is
 * stored as tiles of 16x8 texels = one 128-byte line each (tile rows 16 bytes apart, tiles row-major):
 * the warp reads slanted footprints, and a line that covers a compact 2-D patch is shared by far fewer
 * workgroup blocks than a 128x1 strip of a row.  16 consecutive texels of a row (x % 16 == 0) stay one
 * aligned 16-byte chunk.  Byte offset of texel (plate, px, py) inside one globe frame: */
#if defined(__HIP__) || defined(__HIPCC_RTC__)
#define BK_LAYOUT_FN static __host__ __device__ inline __attribute__((always_inline))
#else
#define BK_LAYOUT_FN static inline
#endif
BK_LAYOUT_FN unsigned int bk_texel_offset(unsigned int gp, unsigned int ph, unsigned int plate, unsigned int px, unsigned int py)
{
    return plate * (gp * ph) + ((py >> 3) * (gp >> 4) + (px >> 4)) * 128u + (py & 7u) * 16u + (px & 15u);
}
/* inverse: offset -> plate, px, py */
BK_LAYOUT_FN void bk_texel_coords(unsigned int gp, unsigned int ph, unsigned int off, unsigned int *plate, unsigned int *px, unsigned int *py)
{
    const unsigned int p = off / (gp * ph), rem = off - p * (gp * ph), tile = rem >> 7, tpr = gp >> 4;
    const unsigned int ty = tile / tpr, tx = tile - ty * tpr;
    *plate = p;
    *px = tx * 16u + (rem & 15u);
    *py = ty * 8u + ((rem >> 4) & 7u);
}

#define BK_ERR_ARITH 1       /* arithmetic on a non-number */
#define BK_ERR_COMPARE 2     /* ordering comparison on non-numbers */
#define BK_ERR_INDEX 4       /* table store out of range / unsupported index */
#define BK_ERR_RESULT 8      /* callback returned a malformed result (status -1, fisheye.c:1565-1584) */
#define BK_ERR_LOOP 16       /* per-pixel iteration budget exceeded */

#endif
