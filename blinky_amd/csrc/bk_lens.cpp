// bk_lens.cpp -- script front-end glue (placeholder until the Lua front-end lands)
#include "bk_internal.h"
namespace bk { struct LensProgram {}; void lensprogram_free(LensProgram *p) { delete p; } }
#define NOTYET(ctx) ((ctx) ? (ctx)->fail(BK_E_STATE, "%s: script front-end not built yet", __func__) : BK_E_INVALID)
extern "C" int bk_load_globe(bk_ctx *ctx, const char *, size_t, const char *) { return NOTYET(ctx); }
extern "C" int bk_load_lens(bk_ctx *ctx, const char *, size_t, const char *) { return NOTYET(ctx); }
extern "C" int bk_get_lens_info(const bk_ctx *, bk_lens_info *) { return BK_E_STATE; }
extern "C" int bk_get_globe(const bk_ctx *ctx, bk_plate plates[BK_MAX_PLATES], int *n)
{
    if (!ctx) return BK_E_INVALID;
    for (int i = 0; i < ctx->numplates; ++i) plates[i] = ctx->plates[i];
    if (n) *n = ctx->numplates;
    return BK_OK;
}
extern "C" int bk_set_globe_plates(bk_ctx *ctx, const bk_plate *plates, int numplates)
{
    if (!ctx || !plates || numplates < 1 || numplates > BK_MAX_PLATES) return BK_E_INVALID;
    for (int i = 0; i < numplates; ++i) ctx->plates[i] = plates[i];
    ctx->numplates = numplates;
    ctx->globe_valid = true;
    return BK_OK;
}
extern "C" int bk_build(bk_ctx *ctx, int *, double *) { return NOTYET(ctx); }
extern "C" int bk_calc_zoom(bk_ctx *ctx, double *) { return NOTYET(ctx); }
