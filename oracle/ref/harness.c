/*
 * harness.c -- ORACLE test infrastructure: oracle/_ref/libref.so
 *
 * Compiles the UNMODIFIED reference translation unit engine/NQ/fisheye.c (by
 * #include, from where it lies under /root/reference -- nothing is copied) and
 * supplies the engine services it links against as stubs, so
 * that the real F_Init -> cmd_* -> F_RenderView -> create_lensmap ->
 * render_lensmap run on chosen inputs.  common/mathlib.c is compiled as-is.
 * Built by oracle/Makefile (target _ref) with the reference's own -DNQ_HACK -DELF.
 */
#include FISHEYE_C      /* -DFISHEYE_C='"/root/reference/engine/NQ/fisheye.c"' */

#include <stdarg.h>
#include <stdint.h>

/* ---- engine data ---------------------------------------------------------------- */
viddef_t vid;
refdef_t r_refdef;
vrect_t scr_vrect;
int sb_lines;
byte *host_basepal;
char com_basedir[MAX_OSPATH];
cmd_source_t cmd_source;
static short little_short(short l) { return l; }
short (*LittleShort)(short l) = little_short;

static int quiet = 1;

/* ---- console / command system ------------------------------------------------- */
#define MAX_CMDS 64
static struct { const char *name; xcommand_t fn; } cmds[MAX_CMDS];
static int ncmds;
static char argbuf[1024];
static char *argv_[16];
static int argc_;

void Con_Printf(const char *fmt, ...)
{
    va_list ap;
    if (quiet) return;
    va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}
void Sys_Error(const char *error, ...)
{
    va_list ap;
    va_start(ap, error); vfprintf(stderr, error, ap); va_end(ap);
    abort();
}
void Cmd_AddCommand(const char *cmd_name, xcommand_t function)
{
    cmds[ncmds].name = cmd_name; cmds[ncmds].fn = function; ncmds++;
}
void Cmd_SetCompletion(const char *cmd_name, cmd_arg_f completion) { (void)cmd_name; (void)completion; }
int Cmd_Argc(void) { return argc_; }
const char *Cmd_Argv(int arg) { return arg < argc_ ? argv_[arg] : ""; }
/* whitespace / double-quote tokeniser, enough for the commands fisheye.c issues */
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
        if (*p == '"') {
            argv_[argc_++] = ++p;
            while (*p && *p != '"') p++;
        } else {
            argv_[argc_++] = p;
            while (*p && *p != ' ' && *p != '\t' && *p != '\n') p++;
        }
        if (*p) *p++ = 0;
    }
    if (!argc_) return;
    for (i = 0; i < ncmds; ++i)
        if (!strcmp(cmds[i].name, argv_[0])) { cmds[i].fn(); return; }
    /* bind/unbind/impulse etc.: not ours */
}
float Q_atof(const char *str) { return (float)atof(str); }
int Q_atoi(const char *str) { return atoi(str); }
void *Z_Malloc(int size) { return calloc(1, (size_t)size); }
void *Hunk_TempAlloc(int size) { return calloc(1, (size_t)size); }
void STree_AllocInit(void) {}
void COM_ScanDir(struct stree_root *root, const char *path, const char *pfx, const char *ext, qboolean stripext)
{ (void)root; (void)path; (void)pfx; (void)ext; (void)stripext; }
/* files the reference writes (f_saveglobe's PCX plates) are kept in memory for the tests */
#define REF_MAX_FILES 8
static struct { char name[64]; unsigned char *data; int len; } ref_files[REF_MAX_FILES];
static int ref_nfiles;
void COM_WriteFile(const char *filename, const void *data, int len)
{
    if (ref_nfiles >= REF_MAX_FILES) return;
    strncpy(ref_files[ref_nfiles].name, filename, sizeof ref_files[0].name - 1);
    ref_files[ref_nfiles].data = (unsigned char *)malloc((size_t)len);
    memcpy(ref_files[ref_nfiles].data, data, (size_t)len);
    ref_files[ref_nfiles].len = len;
    ++ref_nfiles;
}

/* ---- renderer hooks -------------------------------------------------------------- */
void R_PushDlights(void) {}
void R_RenderView(void) {}                      /* plates are filled by ref_run() below */
void R_ViewChanged(vrect_t *pvrect, int lineadj, float aspect) { (void)pvrect; (void)lineadj; (void)aspect; }
void R_SetVrect(const vrect_t *pvrectin, vrect_t *pvrect, int lineadj)
{
    (void)lineadj;
    *pvrect = *pvrectin;                        /* full rect at (0,0): SURVEY.md 8(d) */
}
void Draw_TileClear(int x, int y, int w, int h)
{
    int r;
    for (r = 0; r < h; ++r) memset(vid.buffer + x + (size_t)(y + r) * vid.rowbytes, 0, (size_t)w);
}
void D_EnableBackBufferAccess(void) {}
void D_DisableBackBufferAccess(void) {}

/* ---- driver ------------------------------------------------------------------------ */
static int inited;
static byte basepal[768];

/* SURVEY.md 8(d) LCG */
static void lcg_fill(byte *dst, size_t n, int plate, int frame)
{
    uint32_t st = 0x9E3779B9u * (uint32_t)(plate + 1 + 6 * frame);
    size_t i;
    for (i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; dst[i] = (byte)(st >> 24); }
}

/*
 * One reference run: "f_globe G; f_lens L; [zoomcmd]" then F_RenderView at W x H.
 *   offsets/tints: W*H outputs (offset = ptr - globe.pixels, 0xFFFFFFFF = NULL)
 *   display[6], *scale, *numplates: reference state after the build
 *   frame (nullable): W*H bytes, render_lensmap() over LCG(frame_index) plates
 *   rubix_on: value of rubix.enabled for that apply; grid = "f_rubixgrid" args or NULL
 * Returns 1 if a lensmap was built (lens+globe valid and calc_zoom ok).
 */
int ref_run(const char *globe_name, const char *lens_name, const char *zoomcmd, int W, int H,
            uint32_t *offsets, uint8_t *tints, int *display, double *scale, int *numplates,
            uint8_t *frame, int frame_index, int rubix_on, const char *rubixgrid, int verbose)
{
    char cmd[256];
    size_t area = (size_t)W * H, i;
    int p, built;

    quiet = !verbose;
    if (!inited) {
        for (p = 0; p < 768; ++p) basepal[p] = (byte)((p * 37) % 256);   /* SURVEY.md 8(d) */
        host_basepal = basepal;
        strcpy(com_basedir, "/ref");
        vid.width = W; vid.height = H; vid.rowbytes = W;
        F_Init();
        inited = 1;
    }
    free(vid.buffer);
    vid.buffer = (pixel_t *)calloc(area, 1);
    vid.width = W; vid.height = H; vid.rowbytes = W; vid.aspect = 1;
    scr_vrect.x = 0; scr_vrect.y = 0; scr_vrect.width = W; scr_vrect.height = H;
    lens_builder.seconds_per_frame = 1e9f;      /* run to completion */

    snprintf(cmd, sizeof cmd, "f_rubixgrid %s", rubixgrid ? rubixgrid : "10 4 1");
    Cmd_ExecuteString(cmd, src_command);
    snprintf(cmd, sizeof cmd, "f_globe %s", globe_name);
    Cmd_ExecuteString(cmd, src_command);
    snprintf(cmd, sizeof cmd, "f_lens %s", lens_name);
    Cmd_ExecuteString(cmd, src_command);        /* runs the lens' onload */
    if (zoomcmd && *zoomcmd) Cmd_ExecuteString(zoomcmd, src_command);
    rubix.enabled = false;

    lens.scale = -1;
    F_RenderView();                             /* build (to completion) + stub plates + apply */
    /* calc_zoom's own verdict (fisheye.c:1378: it fails on "scale <= 0" - a NaN scale passes and builds an all-NULL table) */
    built = lens.valid && globe.valid && !(lens.scale <= 0) && !lens_builder.working;

    for (i = 0; i < area; ++i)
        offsets[i] = lens.pixels[i] ? (uint32_t)(lens.pixels[i] - globe.pixels) : 0xFFFFFFFFu;
    memcpy(tints, lens.pixel_tints, area);
    for (p = 0; p < MAX_PLATES; ++p) display[p] = p < globe.numplates ? globe.plates[p].display : 0;
    *scale = lens.scale;
    *numplates = globe.numplates;

    if (frame) {
        size_t ps2 = (size_t)globe.platesize * globe.platesize;
        for (p = 0; p < globe.numplates; ++p) lcg_fill(globe.pixels + ps2 * p, ps2, p, frame_index);
        memset(vid.buffer, 0, area);
        rubix.enabled = rubix_on ? true : false;
        render_lensmap();                       /* the reference's own apply */
        rubix.enabled = false;
        memcpy(frame, vid.buffer, area);
    }
    return built;
}

/* Time the reference's own render_lensmap() (fisheye.c:2406-2424) on the state the last ref_run
 * left behind: LCG plates, rubix off.  Returns total seconds over `reps` calls; *best_ms = fastest. */
#include <time.h>
double ref_time_apply(int reps, double *best_ms)
{
    size_t ps2 = (size_t)globe.platesize * globe.platesize;
    struct timespec a, b;
    double total = 0, best = 1e30;
    int p, i;
    for (p = 0; p < globe.numplates; ++p) lcg_fill(globe.pixels + ps2 * p, ps2, p, 0);
    rubix.enabled = false;
    for (i = 0; i < reps; ++i) {
        double dt;
        clock_gettime(CLOCK_MONOTONIC, &a);
        render_lensmap();
        clock_gettime(CLOCK_MONOTONIC, &b);
        dt = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        total += dt;
        if (dt < best) best = dt;
    }
    if (best_ms) *best_ms = best * 1e3;
    return total;
}

/* "f_saveglobe <name> <with_margins>" on the state the last ref_run left behind, plates = LCG(frame_index):
 * runs the reference's own cmd_saveglobe + save_globe + WritePCXplate (fisheye.c:1120-1136, 1396-1484) and
 * returns the bytes it handed to COM_WriteFile for plate `plate` (length, or -1; name_out gets the file name). */
int ref_saveglobe(const char *name, int with_margins, int frame_index, int plate, unsigned char *out, int cap, char *name_out)
{
    char cmd[128];
    size_t ps2 = (size_t)globe.platesize * globe.platesize;
    int p, i, len = -1;
    for (p = 0; p < globe.numplates; ++p) lcg_fill(globe.pixels + ps2 * p, ps2, p, frame_index);
    for (i = 0; i < ref_nfiles; ++i) free(ref_files[i].data);
    ref_nfiles = 0;
    snprintf(cmd, sizeof cmd, "f_saveglobe %s %d", name, with_margins);
    Cmd_ExecuteString(cmd, src_command);
    if (!globe.save.should) return -1;
    save_globe();
    if (plate >= 0 && plate < ref_nfiles) {
        len = ref_files[plate].len;
        if (len <= cap) memcpy(out, ref_files[plate].data, (size_t)len);
        if (name_out) strcpy(name_out, ref_files[plate].name);
    }
    return len;
}

/* the reference's rubix palette LUTs (create_palmap ran in F_Init) */
void ref_palettes(uint8_t out[6][256])
{
    int p;
    for (p = 0; p < MAX_PLATES; ++p) memcpy(out[p], globe.plates[p].palette, 256);
}

/* the same for any base palette: the reference's own create_palmap (fisheye.c:835-908) on `pal` (768 bytes) */
void ref_palettes_of(const uint8_t *pal, uint8_t out[6][256])
{
    byte *saved = host_basepal;
    int p;
    host_basepal = (byte *)pal;
    create_palmap();
    for (p = 0; p < MAX_PLATES; ++p) memcpy(out[p], globe.plates[p].palette, 256);
    host_basepal = saved;
    create_palmap();
}
