/*
 * engine_stub.c -- TEST INFRASTRUCTURE: a stand-in for the TyrQuake services that
 * blinky_amd/host/fisheye_hip.c links against (see blinky_amd/host/engine_iface.h), plus a small
 * driver API for the Python tests.  The "scene renderer" R_RenderView paints the view rectangle
 * with the SURVEY.md 8(d) LCG stream of the plate being rendered.
 */
#include "../../blinky_amd/host/engine_iface.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* fisheye_hip.c's public face (engine/include/fisheye.h) */
extern qboolean fisheye_enabled;
extern double fisheye_plate_fov;
void F_Init(void);
void F_Shutdown(void);
void F_RenderView(void);
void F_WriteConfig(FILE *f);

viddef_t vid;
refdef_t r_refdef;
vrect_t scr_vrect;
int sb_lines;
byte *host_basepal;
#ifdef BLINKY_IN_ENGINE
char com_basedir[MAX_OSPATH];                  /* (the engine's own declaration: include/common.h:201) */
#else
char com_basedir[1024];
#endif

static byte basepal[768];
static char console[1 << 16];
static size_t console_len;
static int plate_order[8], plate_count, plate_next, frame_index;
static double plate_fovs[8];
static int bg_value;

/* ---- console / commands ------------------------------------------------------------------------ */
void Con_Printf(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (console_len < sizeof console - 1)
        console_len += (size_t)vsnprintf(console + console_len, sizeof console - console_len, fmt, ap);
    if (console_len >= sizeof console) console_len = sizeof console - 1;
    va_end(ap);
}
#define MAX_CMDS 64
static struct { const char *name; xcommand_t fn; } cmds[MAX_CMDS];
static int ncmds, argc_;
static char argbuf[1024], *argv_[16];
void Cmd_AddCommand(const char *cmd_name, xcommand_t function) { cmds[ncmds].name = cmd_name; cmds[ncmds].fn = function; ncmds++; }
/* tab completion: the registered callbacks are kept and can be invoked by the tests; COM_ScanDir lists <basedir>/<path
 * without the leading "../"> like the engine's (which scans relative to the game directory, a child of the base dir) */
static struct { const char *name; cmd_arg_f fn; } completions[8];
static int ncompletions;
static char scan_result[4096];
void Cmd_SetCompletion(const char *cmd_name, cmd_arg_f completion)
{
    if (ncompletions < 8) { completions[ncompletions].name = cmd_name; completions[ncompletions].fn = completion; ncompletions++; }
}
void *Z_Malloc(int size) { return calloc(1, (size_t)size); }
void STree_AllocInit(void) {}
#include <dirent.h>
static int cmp_str(const void *a, const void *b) { return strcmp(*(const char *const *)a, *(const char *const *)b); }
void COM_ScanDir(struct stree_root *root, const char *path, const char *pfx, const char *ext, qboolean stripext)
{
    char dirpath[1200], *names[256];
    DIR *d;
    struct dirent *e;
    int n = 0, i;
    size_t el = strlen(ext), pl = pfx ? strlen(pfx) : 0;
    snprintf(dirpath, sizeof dirpath, "%s/%s", com_basedir, strncmp(path, "../", 3) ? path : path + 3);
    scan_result[0] = 0;
    d = opendir(dirpath);
    if (!d) return;
    while ((e = readdir(d)) && n < 256) {
        size_t l = strlen(e->d_name);
        if (l <= el || strcmp(e->d_name + l - el, ext)) continue;
        if (pl && strncmp(e->d_name, pfx, pl)) continue;
        names[n] = strdup(e->d_name);
        if (stripext) names[n][l - el] = 0;
        n++;
    }
    closedir(d);
    qsort(names, (size_t)n, sizeof names[0], cmp_str);
    for (i = 0; i < n; ++i) {
        strncat(scan_result, names[i], sizeof scan_result - strlen(scan_result) - 2);
        strcat(scan_result, " ");
        if (strlen(names[i]) > root->maxlen) root->maxlen = (unsigned)strlen(names[i]);
        root->entries++;
        free(names[i]);
    }
}
/* the completions of `cmd <arg>`: space-separated names, -1 if the command has no completion callback */
int hosttest_complete(const char *cmd, const char *arg, char *out, int cap)
{
    int i;
    for (i = 0; i < ncompletions; ++i)
        if (!strcmp(completions[i].name, cmd)) {
            struct stree_root *r = completions[i].fn(arg);
            int n = r ? (int)r->entries : -1;
            snprintf(out, (size_t)cap, "%s", scan_result);
            free(r);
            return n;
        }
    return -1;
}
int Cmd_Argc(void) { return argc_; }
const char *Cmd_Argv(int arg) { return arg < argc_ ? argv_[arg] : ""; }
void Cmd_ExecuteString(const char *text, cmd_source_t src)
{
    char *p;
    int i;
    (void)src;
    strncpy(argbuf, text, sizeof argbuf - 1);
    argbuf[sizeof argbuf - 1] = 0;
    argc_ = 0;
    p = argbuf;
    while (*p && argc_ < 16) {
        while (*p == ' ' || *p == '\t' || *p == '\n') p++;
        if (!*p) break;
        if (*p == '"') { argv_[argc_++] = ++p; while (*p && *p != '"') p++; }
        else { argv_[argc_++] = p; while (*p && *p != ' ' && *p != '\t' && *p != '\n') p++; }
        if (*p) *p++ = 0;
    }
    if (!argc_) return;
    for (i = 0; i < ncmds; ++i)
        if (!strcmp(cmds[i].name, argv_[0])) { cmds[i].fn(); return; }
    Con_Printf("[engine] %s\n", text);            /* bind / unbind / impulse ...: record them */
}
float Q_atof(const char *str) { return (float)atof(str); }
int Q_atoi(const char *str) { return atoi(str); }

/* ---- math (oracle/ref/hosttest_ref.c links the reference's own common/mathlib.c instead) ---------------- */
#ifndef HOSTTEST_WITH_REFERENCE
void VectorMA(const vec3_t veca, const float scale, const vec3_t vecb, vec3_t vecc)
{
    vecc[0] = veca[0] + scale * vecb[0];
    vecc[1] = veca[1] + scale * vecb[1];
    vecc[2] = veca[2] + scale * vecb[2];
}
void AngleVectors(const vec3_t angles, vec3_t forward, vec3_t right, vec3_t up)
{
    /* Quake convention: angles = (pitch, yaw, roll) in degrees */
    float sy = sinf(angles[1] * (float)(M_PI / 180)), cy = cosf(angles[1] * (float)(M_PI / 180));
    float sp = sinf(angles[0] * (float)(M_PI / 180)), cp = cosf(angles[0] * (float)(M_PI / 180));
    float sr = sinf(angles[2] * (float)(M_PI / 180)), cr = cosf(angles[2] * (float)(M_PI / 180));
    forward[0] = cp * cy; forward[1] = cp * sy; forward[2] = -sp;
    right[0] = -1 * sr * sp * cy + -1 * cr * -sy; right[1] = -1 * sr * sp * sy + -1 * cr * cy; right[2] = -1 * sr * cp;
    up[0] = cr * sp * cy + -sr * -sy; up[1] = cr * sp * sy + -sr * cy; up[2] = cr * cp;
}
#endif

/* ---- zone / filesystem: files the host layer writes (f_saveglobe) are kept in memory ------------------- */
void *Hunk_TempAlloc(int size)
{
    static void *last;
    free(last);
    last = calloc(1, (size_t)size);
    return last;
}
#define MAX_FILES 8
static struct { char name[64]; unsigned char *data; int len; } files[MAX_FILES];
static int nfiles;
void COM_WriteFile(const char *filename, const void *data, int len)
{
    if (nfiles >= MAX_FILES) return;
    snprintf(files[nfiles].name, sizeof files[0].name, "%s", filename);
    files[nfiles].data = (unsigned char *)malloc((size_t)len);
    memcpy(files[nfiles].data, data, (size_t)len);
    files[nfiles].len = len;
    ++nfiles;
}
void D_EnableBackBufferAccess(void) {}
void D_DisableBackBufferAccess(void) {}
int hosttest_num_files(void) { return nfiles; }
int hosttest_file(int i, char *name_out, unsigned char *out, int cap)
{
    if (i < 0 || i >= nfiles) return -1;
    strcpy(name_out, files[i].name);
    if (files[i].len <= cap) memcpy(out, files[i].data, (size_t)files[i].len);
    return files[i].len;
}
void hosttest_clear_files(void)
{
    int i;
    for (i = 0; i < nfiles; ++i) free(files[i].data);
    nfiles = 0;
}

/* ---- renderer hooks ---------------------------------------------------------------------------------- */
void R_PushDlights(void) {}
void R_ViewChanged(vrect_t *pvrect, int lineadj, float aspect) { (void)pvrect; (void)lineadj; (void)aspect; }
void R_SetVrect(const vrect_t *pvrectin, vrect_t *pvrect, int lineadj) { (void)pvrectin; (void)pvrect; (void)lineadj; }
void Draw_TileClear(int x, int y, int w, int h)
{
    int r;
    for (r = 0; r < h; ++r) memset(vid.buffer + x + (size_t)(y + r) * vid.rowbytes, bg_value, (size_t)w);
}
void R_RenderView(void)
{
    /* paint the square plate view with the LCG stream of the plate the host is rendering */
    int ps = scr_vrect.width < scr_vrect.height ? scr_vrect.width : scr_vrect.height;
    int p = plate_next < plate_count ? plate_order[plate_next] : 0;
    uint32_t st = 0x9E3779B9u * (uint32_t)(p + 1 + 6 * frame_index);
    int x, y;
    if (plate_next < 8) plate_fovs[plate_next] = fisheye_plate_fov;
    plate_next++;
    for (y = 0; y < ps; ++y) {
        byte *row = vid.buffer + scr_vrect.x + (size_t)(scr_vrect.y + y) * vid.rowbytes;
        for (x = 0; x < ps; ++x) { st = st * 1664525u + 1013904223u; row[x] = (byte)(st >> 24); }
    }
}

/* ---- driver API for the tests -------------------------------------------------------------------------- */
int hosttest_init(const char *basedir)
{
    int i;
    for (i = 0; i < 768; ++i) basepal[i] = (byte)((i * 37) % 256);
    host_basepal = basepal;
    snprintf(com_basedir, sizeof com_basedir, "%s", basedir);
    console_len = 0; console[0] = 0;
    F_Init();
    return fisheye_enabled;
}
void hosttest_resize(int W, int H, int x0, int y0, int extra_pitch)
{
    free(vid.buffer);
    vid.width = W + x0 + 3; vid.height = H + y0 + 2;
    vid.rowbytes = vid.width + extra_pitch;
    vid.buffer = (pixel_t *)calloc((size_t)vid.rowbytes * vid.height, 1);
    vid.aspect = 1;
    scr_vrect.x = x0; scr_vrect.y = y0; scr_vrect.width = W; scr_vrect.height = H;
}
void hosttest_cmd(const char *text) { Cmd_ExecuteString(text, src_command); }
/* one frame: plates are painted in `order` (the display[] flags, ascending); returns how many the host rendered */
int hosttest_frame(const int *order, int n, int frame, int bg, uint8_t *out)
{
    int i;
    plate_count = n; plate_next = 0; frame_index = frame; bg_value = bg;
    for (i = 0; i < n && i < 8; ++i) plate_order[i] = order[i];
    memset(vid.buffer, bg, (size_t)vid.rowbytes * vid.height);
    F_RenderView();
    if (out) memcpy(out, vid.buffer, (size_t)vid.rowbytes * vid.height);
    return plate_next;
}
int hosttest_rowbytes(void) { return vid.rowbytes; }
int hosttest_vidheight(void) { return vid.height; }
double hosttest_plate_fov(int i) { return plate_fovs[i]; }
const char *hosttest_console(void) { return console; }
void hosttest_console_clear(void) { console_len = 0; console[0] = 0; }
int hosttest_writeconfig(char *buf, int cap)
{
    FILE *f = tmpfile();
    size_t n;
    if (!f) return -1;
    F_WriteConfig(f);
    fflush(f);
    rewind(f);
    n = fread(buf, 1, (size_t)cap - 1, f);
    buf[n] = 0;
    fclose(f);
    return (int)n;
}
void hosttest_shutdown(void) { F_Shutdown(); }
