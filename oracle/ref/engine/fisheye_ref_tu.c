/*
 * fisheye_ref_tu.c -- ORACLE test infrastructure: the UNMODIFIED reference translation unit engine/NQ/fisheye.c as the
 * reference engine of tests/test_engine_dropin.py links it (compiled by #include from where it lies under /root/reference,
 * its Lua through oracle/ref's C-API shim with the hand-C scripts: no Lua VM exists in this image).
 *
 * One thing is added around it: the reference spreads a lensmap build over frames by wall-clock time
 * (lens_builder.seconds_per_frame = 1/60 s, fisheye.c:645, 813-826), so which frame first shows a new lens depends on how
 * fast the machine is.  For frame-by-frame comparisons every build runs to completion: F_Init is the
 * reference's F_Init followed by that one assignment.
 */
#define F_Init F_Init_as_the_reference_wrote_it
#include FISHEYE_C      /* -DFISHEYE_C='"/root/reference/engine/NQ/fisheye.c"' */
#undef F_Init

void F_Init(void)
{
    F_Init_as_the_reference_wrote_it();
    lens_builder.seconds_per_frame = 1e9f;
}
