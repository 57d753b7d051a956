/*
 * hosttest_ref.c -- ORACLE test infrastructure: oracle/_ref/libhosttest_ref.so
 *
 * The engine stand-in of tests/host/engine_stub.c - the driver API the host-layer tests speak (hosttest_init / _cmd /
 * _resize / _frame / _writeconfig / _console) - linked to the UNMODIFIED reference translation unit engine/NQ/fisheye.c
 * instead of blinky_amd/host/fisheye_hip.c.  The reference is compiled by #include from where it lies under /root/reference
 * (nothing is copied), common/mathlib.c as it is, its Lua through oracle/ref's shim with the hand-C scripts.  With it the
 * same sequence of console commands, resizes and frames can be put to the reference's F_RenderView and to the product's, and
 * everything a user can observe compared: the frame, the plates the engine was asked to render and their fov, the console text
 * and the config (tests/test_host_differential_gpu.py).  Built by oracle/Makefile (target _ref).
 */
#include FISHEYE_C      /* -DFISHEYE_C='"/root/reference/engine/NQ/fisheye.c"' */

cmd_source_t cmd_source;
static short little_short(short l) { return l; }
short (*LittleShort)(short l) = little_short;
void Sys_Error(const char *error, ...) { (void)error; abort(); }

#define HOSTTEST_WITH_REFERENCE 1
#include "../../tests/host/engine_stub.c"

/* the reference spreads a build over frames (lens_builder.seconds_per_frame, fisheye.c:303-330): run every build to completion, as
 * SURVEY.md A.6 prescribes for comparisons, so that a frame shows the whole new lensmap */
void hosttest_ref_build_to_completion(void) { lens_builder.seconds_per_frame = 1e9f; }
