/*
 * oracle_cli.c -- CPU ORACLE command line (test infrastructure only).
 *   oracle_cli <globe> <lens> <zoomcmd|-> <W> <H> [apply_reps]
 * prints scale, display flags, non-NULL count, FNV-1a-64 of offsets / tints
 * (the hashing convention SURVEY.md Appendix C describes: little-endian uint32
 * offsets = ptr - globe.pixels, 0xFFFFFFFF for NULL, row-major; tints as bytes)
 * and, with apply_reps, the best-of-N time of ok_apply on LCG globe faces plus
 * the frame hash.  NOTE: the hash VALUES printed in Appendix C are not
 * reproduced by the unmodified reference compiled here (oracle/_ref) - scale,
 * display flags and non-NULL counts are; the survey itself says "re-derive
 * before trusting".  What pins this oracle is tests/golden/lensmaps.json,
 * recorded from oracle/_ref by tests/golden/make_golden.py.
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    ok_state s;
    int W, H, i, reps = 0;
    size_t area, nn = 0, k;
    double t0, t1;
    if (argc < 6) { fprintf(stderr, "usage: %s globe lens zoomcmd|- W H [apply_reps]\n", argv[0]); return 2; }
    W = atoi(argv[4]); H = atoi(argv[5]);
    if (argc > 6) reps = atoi(argv[6]);
    if (!ok_configure(&s, argv[1], argv[2], strcmp(argv[3], "-") ? argv[3] : NULL, W, H)) {
        fprintf(stderr, "unknown globe/lens\n"); return 1;
    }
    area = (size_t)W * H;
    s.offsets = (uint32_t *)malloc(area * 4);
    s.tints = (uint8_t *)malloc(area);
    t0 = now();
    if (!ok_create_lensmap(&s)) { fprintf(stderr, "lensmap build failed\n"); return 1; }
    t1 = now();
    for (k = 0; k < area; ++k) nn += s.offsets[k] != OK_NULL_OFFSET;
    printf("config %dx%d %s %s ps=%d scale=%.17g map=%s\n", W, H, argv[1], argv[2], s.platesize, s.scale,
           s.map_type == OK_MAP_INVERSE ? "inverse" : "forward");
    printf("display");
    for (i = 0; i < s.numplates; ++i) printf(" %d", s.plates[i].display);
    printf("\nnonnull %zu/%zu\n", nn, area);
    printf("fnv_offsets %016llx\nfnv_tints %016llx\n",
           (unsigned long long)ok_fnv1a64(s.offsets, area * 4), (unsigned long long)ok_fnv1a64(s.tints, area));
    printf("build_ms %.3f\n", (t1 - t0) * 1e3);
    if (reps > 0) {
        size_t ps2 = (size_t)s.platesize * s.platesize;
        uint8_t *globe = (uint8_t *)malloc(ps2 * OK_MAX_PLATES);
        uint8_t *dst = (uint8_t *)calloc(area, 1);
        double best = 1e30;
        for (i = 0; i < s.numplates; ++i) ok_lcg_fill_plate(globe + ps2 * i, ps2, i, 0);
        for (i = 0; i < reps; ++i) {
            t0 = now(); ok_apply(&s, globe, dst, W, 0, 0, 0); t1 = now();
            if (t1 - t0 < best) best = t1 - t0;
        }
        printf("apply_ms %.4f  Mpx/s %.1f\nfnv_frame %016llx\n", best * 1e3, area / best / 1e6,
               (unsigned long long)ok_fnv1a64(dst, area));
    }
    return 0;
}
