/* multi_test.c -- TEST: a plain C host (no HIP, no Python) spreading the warp over several "devices" through
 * bk_multi_* (include/blinky_hip.h), the way blinky_amd/host/fisheye_hip.c does for the engine.
 *   multi_test <globe.lua> <lens.lua> <W> <H> <nframes> <dev0,dev1,...> <outfile>
 * Builds the lensmap stripe by stripe, fills LCG plates, then
 *   (1) bk_multi_apply          -> host frame 0
 *   (2) bk_multi_apply_stripes + bk_multi_gather            -> all frames on device 0
 *   (3) bk_multi_apply_stripes + bk_multi_exchange_rotating -> frame f on device f % N
 * and writes the frames of (1), (2), (3) to <outfile> for the Python test to compare with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/blinky_hip.h"

static char *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    char *b;
    long len;
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); len = ftell(f); fseek(f, 0, SEEK_SET);
    b = (char *)malloc((size_t)len + 1);
    if (fread(b, 1, (size_t)len, f) != (size_t)len) exit(2);
    fclose(f);
    b[len] = 0; *n = (size_t)len;
    return b;
}

#define CK(call) do { int rc_ = (call); if (rc_ != BK_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bk_multi_last_error(m)); return 1; } } while (0)

int main(int argc, char **argv)
{
    int devs[16], ndev = 0, W, H, F, i, f, p, display[BK_MAX_PLATES];
    size_t gl, ll, frame_bytes;
    char *gsrc, *lsrc, *tok;
    double scale;
    bk_multi *m;
    bk_lens_info info;
    void *stripes[16], *frames[16], *all;
    unsigned char *host;
    FILE *out;
    if (argc != 8) { fprintf(stderr, "usage: multi_test globe.lua lens.lua W H nframes devs out\n"); return 2; }
    gsrc = slurp(argv[1], &gl); lsrc = slurp(argv[2], &ll);
    W = atoi(argv[3]); H = atoi(argv[4]); F = atoi(argv[5]);
    for (tok = strtok(argv[6], ","); tok && ndev < 16; tok = strtok(NULL, ",")) devs[ndev++] = atoi(tok);
    frame_bytes = (size_t)W * H;

    m = bk_create_multi(ndev, devs);
    if (!m) { fprintf(stderr, "bk_create_multi: %s\n", bk_multi_last_error(NULL)); return 1; }
    CK(bk_multi_set_frames(m, F));
    CK(bk_multi_load_globe(m, gsrc, gl, "globe.lua"));
    CK(bk_multi_load_lens(m, lsrc, ll, "lens.lua"));
    bk_get_lens_info(bk_multi_ctx(m, 0), &info);            /* the lens' onload zoom, as cmd_lens does (fisheye.c:1087-1102) */
    if (!strncmp(info.onload, "f_fov", 5)) CK(bk_multi_set_zoom(m, BK_ZOOM_FOV, atoi(info.onload + 5)));
    else if (!strncmp(info.onload, "f_contain", 9)) CK(bk_multi_set_zoom(m, BK_ZOOM_CONTAIN, 0));
    else if (!strncmp(info.onload, "f_cover", 7)) CK(bk_multi_set_zoom(m, BK_ZOOM_COVER, 0));
    CK(bk_multi_resize(m, W, H));
    CK(bk_multi_build(m, display, &scale));
    printf("ranks %d rccl %d scale %.17g display %d%d%d%d%d%d\n", bk_multi_size(m), bk_multi_uses_rccl(m), scale,
           display[0], display[1], display[2], display[3], display[4], display[5]);
    for (f = 0; f < F; ++f)
        for (p = 0; p < BK_MAX_PLATES; ++p) CK(bk_multi_fill_plate_lcg(m, f, p, (uint32_t)f));

    out = fopen(argv[7], "wb");
    host = (unsigned char *)calloc(frame_bytes, 1);
    /* (1) host frame */
    CK(bk_multi_apply(m, 0, host, W, 0, 0, 0, NULL));
    fwrite(host, 1, frame_bytes, out);

    for (i = 0; i < ndev; ++i) {
        int r0, r1;
        bk_comm_stripe(bk_multi_comm(m, i), i, &r0, &r1);
        stripes[i] = bk_dev_alloc(bk_multi_ctx(m, i), (size_t)F * (size_t)(r1 - r0) * W);
        frames[i] = bk_dev_alloc(bk_multi_ctx(m, i), (size_t)((F + ndev - 1) / ndev) * frame_bytes);
        if (!stripes[i] || !frames[i]) { fprintf(stderr, "bk_dev_alloc failed\n"); return 1; }
    }
    all = bk_dev_alloc(bk_multi_ctx(m, 0), (size_t)F * frame_bytes);
    /* (2) every frame gathered onto rank 0 */
    CK(bk_multi_apply_stripes(m, 0, F, stripes, 0, NULL));
    CK(bk_multi_gather(m, stripes, F, 0, all, frame_bytes, 0));
    CK(bk_multi_synchronize(m));
    for (f = 0; f < F; ++f) {
        if (bk_dev_read(bk_multi_ctx(m, 0), host, (unsigned char *)all + (size_t)f * frame_bytes, frame_bytes) != BK_OK) return 1;
        fwrite(host, 1, frame_bytes, out);
    }
    /* (3) frame f reassembled on rank f % N */
    CK(bk_multi_wait(m, 0));                                   /* the stripe buffers are about to be overwritten */
    CK(bk_multi_apply_stripes(m, 0, F, stripes, 0, NULL));
    CK(bk_multi_exchange_rotating(m, stripes, F, frames, frame_bytes, 1));
    CK(bk_multi_synchronize(m));
    for (f = 0; f < F; ++f) {
        if (bk_dev_read(bk_multi_ctx(m, f % ndev), host, (unsigned char *)frames[f % ndev] + (size_t)(f / ndev) * frame_bytes, frame_bytes) != BK_OK) return 1;
        fwrite(host, 1, frame_bytes, out);
    }
    fclose(out);
    for (i = 0; i < ndev; ++i) { bk_dev_free(bk_multi_ctx(m, i), stripes[i]); bk_dev_free(bk_multi_ctx(m, i), frames[i]); }
    bk_dev_free(bk_multi_ctx(m, 0), all);
    bk_destroy_multi(m);
    printf("ok\n");
    return 0;
}
