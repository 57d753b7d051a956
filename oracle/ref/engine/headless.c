/*
 * headless.c -- ORACLE test infrastructure: a video + input driver without a display for the reference's
 * TyrQuake engine (oracle/ref/engine/Makefile links it where the reference's Makefile would link vid_x.c / in_x11.c;
 * its own vid_null.c / in_null.c no longer compile against the current headers).
 *
 * The driver is the same object file in both engines of tests/test_engine_dropin.py - the reference's, with the unmodified
 * engine/NQ/fisheye.c, and the product's, with blinky_amd/host/fisheye_hip.c in its place - so everything either one
 * shows on "screen" can be compared: VID_Update appends one line per presented frame to $BLINKY_HEADLESS_LOG
 * ("frame <n> <width> <height> <fnv-1a-64 of the visible rows>").
 *
 *   BLINKY_HEADLESS_SIZE=WxH    frame size (default 320x200, the engine's base size)
 *   console command             headless_size <width> <height>: a resize while the engine runs
 *   BLINKY_HEADLESS_LOG=path    per-frame hashes (appended)
 *   BLINKY_HEADLESS_DUMP=dir    also write every presented frame as dir/frame%04d.raw (W*H bytes)
 *   BLINKY_HEADLESS_TIMES=path  wall-clock milliseconds each presented frame took, one per line (tools/engine_fps.py)
 *   BLINKY_HEADLESS_FRAMES=N    leave through Sys_Quit after N presented frames (a script that never reaches "quit" still ends)
 *
 * Interfaces implemented: include/vid.h:104-146 (VID_*), include/input.h:42-61 (IN_*), include/sys.h:73,
 * include/d_iface.h:131-134.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "quakedef.h"
#include "cmd.h"
#include "console.h"
#include "d_local.h"
#include "host.h"
#include "input.h"
#include "sys.h"
#include "vid.h"

viddef_t vid;
unsigned short d_8to16table[256];
unsigned d_8to24table[256];
int vid_modenum = VID_MODE_NONE;
cvar_t _windowed_mouse = { "_windowed_mouse", "0", true };

static byte *frame_buffer;
static short *z_buffer;
static byte *surface_cache;
static int frames_presented;
static struct timespec frame_started;

static void set_size(int width, int height)
{
    int cache_bytes = D_SurfaceCacheForRes(width, height);
    free(frame_buffer);
    free(z_buffer);
    if (surface_cache) D_FlushCaches();
    free(surface_cache);
    frame_buffer = (byte *)calloc((size_t)width * height, 1);
    z_buffer = (short *)calloc((size_t)width * height, sizeof(short));
    surface_cache = (byte *)calloc((size_t)cache_bytes, 1);
    if (!frame_buffer || !z_buffer || !surface_cache) Sys_Error("headless video: out of memory for %dx%d", width, height);
    vid.width = vid.conwidth = width;
    vid.height = vid.conheight = height;
    vid.maxwarpwidth = WARP_WIDTH;
    vid.maxwarpheight = WARP_HEIGHT;
    vid.aspect = ((float)height / (float)width) * (320.0f / 240.0f);
    vid.numpages = 1;
    vid.buffer = vid.conbuffer = vid.direct = frame_buffer;
    vid.rowbytes = vid.conrowbytes = width;
    vid.recalc_refdef = 1;
    d_pzbuffer = z_buffer;
    D_InitCaches(surface_cache, cache_bytes);
}

/* "headless_size <width> <height>": what a window resize or a video mode change is to the engine - new buffers, vid.recalc_refdef */
static void headless_size_f(void)
{
    int width, height;
    if (Cmd_Argc() != 3) { Con_Printf("headless_size <width> <height>\n"); return; }
    width = atoi(Cmd_Argv(1));
    height = atoi(Cmd_Argv(2));
    if (width < 320 || height < 200 || width > MAXWIDTH || height > MAXHEIGHT) { Con_Printf("headless_size: out of range\n"); return; }
    set_size(width, height);
}

void VID_Init(const byte *palette)
{
    int width = 320, height = 200;
    const char *size = getenv("BLINKY_HEADLESS_SIZE");
    if (size && sscanf(size, "%dx%d", &width, &height) != 2) Sys_Error("BLINKY_HEADLESS_SIZE must be WxH");
    if (width < 320 || height < 200 || width > MAXWIDTH || height > MAXHEIGHT)
        Sys_Error("headless video: %dx%d is outside the software renderer's 320x200 .. %dx%d", width, height, MAXWIDTH, MAXHEIGHT);
    vid.colormap = host_colormap;
    vid.fullbright = 256 - LittleLong(*((int *)vid.colormap + 2048));
    set_size(width, height);
    VID_SetPalette(palette);
    Cmd_AddCommand("headless_size", headless_size_f);
}

void VID_Shutdown(void) {}
void VID_SetPalette(const byte *palette) { (void)palette; }
void VID_ShiftPalette(const byte *palette) { (void)palette; }
qboolean VID_SetMode(const qvidmode_t *mode, const byte *palette) { (void)mode; (void)palette; return true; }
qboolean VID_CheckAdequateMem(int width, int height) { (void)width; (void)height; return true; }
qboolean VID_IsFullScreen(void) { return false; }
void VID_LockBuffer(void) {}
void VID_UnlockBuffer(void) {}

void VID_Update(vrect_t *rects)
{
    const char *log = getenv("BLINKY_HEADLESS_LOG"), *dump = getenv("BLINKY_HEADLESS_DUMP");
    uint64_t h = 0xcbf29ce484222325ull;
    int x, y;
    (void)rects;
    for (y = 0; y < vid.height; ++y)
        for (x = 0; x < vid.width; ++x) h = (h ^ vid.buffer[(size_t)y * vid.rowbytes + x]) * 0x100000001b3ull;
    if (log) {
        FILE *f = fopen(log, "a");
        if (f) {
            fprintf(f, "frame %d %d %d %016llx\n", frames_presented, vid.width, vid.height, (unsigned long long)h);
            fclose(f);
        }
    }
    if (dump) {
        char path[1024];
        FILE *f;
        snprintf(path, sizeof path, "%s/frame%04d.raw", dump, frames_presented);
        f = fopen(path, "wb");
        if (f) {
            for (y = 0; y < vid.height; ++y) fwrite(vid.buffer + (size_t)y * vid.rowbytes, 1, (size_t)vid.width, f);
            fclose(f);
        }
    }
    {
        /* the engine runs at most 72 frames a second (Host_FilterTime, NQ/host.c:518): what a frame COSTS is the time from its first
         * call into a driver - Sys_SendKeyEvents, right after the filter let it through (NQ/host.c, _Host_Frame) - to its presentation */
        const char *times = getenv("BLINKY_HEADLESS_TIMES");
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (times && frame_started.tv_sec) {
            FILE *f = fopen(times, "a");
            if (f) {
                fprintf(f, "%.3f\n", (now.tv_sec - frame_started.tv_sec) * 1e3 + (now.tv_nsec - frame_started.tv_nsec) * 1e-6);
                fclose(f);
            }
        }
    }
    ++frames_presented;
    {
        const char *limit = getenv("BLINKY_HEADLESS_FRAMES");
        if (limit && frames_presented >= atoi(limit)) Sys_Quit();
    }
}

void D_BeginDirectRect(int x, int y, const byte *pbitmap, int width, int height) { (void)x; (void)y; (void)pbitmap; (void)width; (void)height; }
void D_EndDirectRect(int x, int y, int width, int height) { (void)x; (void)y; (void)width; (void)height; }

/* no keyboard, no mouse: the console script (quake.rc, +commands) is the only input */
void Sys_SendKeyEvents(void) { clock_gettime(CLOCK_MONOTONIC, &frame_started); }
void IN_Init(void) {}
void IN_Shutdown(void) {}
void IN_Commands(void) {}
void IN_Move(usercmd_t *cmd) { (void)cmd; }
void IN_ModeChanged(void) {}
void IN_ClearStates(void) {}
void IN_Accumulate(void) {}
void IN_ProcessEvents(void) {}
