// bk_emit.h -- Lua AST -> HIP C++ for the lensmap build kernels (see bk_emit.cpp).
#pragma once

#include <string>

#include "bk_lua.h"

namespace bk {

struct EmitRequest {
    bklua::Interp *interp = nullptr;
    bklua::Value lens_inverse;      // NIL when the lens has none
    bklua::Value lens_forward;
    bklua::Value globe_plate;       // NIL = argmax of dot products (fisheye.c:2035-2047)
};

// Generates the complete translation unit handed to hiprtc: bkm.h + tagged-value runtime +
// the script functions + the generic build kernels.  Throws bklua::LuaError with a message
// naming the construct when a script uses something the device compiler does not support.
std::string emit_build_source(const EmitRequest &req);

// Does a callback READ a script global that callbacks assign before assigning it itself on that path - i.e. can a value travel
// from one pixel's evaluation to the next (eckert4's per-row cache; a counter)?  Conservative definite-assignment walk over the
// callbacks and the script functions they call (assignments inside callees are not credited to the caller).  `which` (nullable)
// names the first such global.  The parallel GPU build gives every pixel the post-load value instead (include/blinky_hip.h: bk_build).
bool callbacks_carry_state(const EmitRequest &req, std::string *which);

// headers the generated unit #includes, embedded at build time (bk_embed.inc) and handed to
// hiprtcCreateProgram: bkm.h, bkm_tables.h, bk_build_params.h, bk_device_rt.h, bk_build_kernels.h
struct EmbeddedHeader { const char *name; const char *text; };
extern const EmbeddedHeader kEmbeddedHeaders[];
extern const int kNumEmbeddedHeaders;

}  // namespace bk
