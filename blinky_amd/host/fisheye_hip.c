/*
 * fisheye_hip.c -- drop-in replacement of engine/NQ/fisheye.c for TyrQuake: same public face
 * (engine/include/fisheye.h:4-9: fisheye_enabled, F_Init, F_Shutdown, F_RenderView,
 * F_WriteConfig, plus fisheye_plate_fov read by common/r_main.c:417-418), same console commands
 * (fisheye.c:651-665) and script locations (fisheye.c:1666, 1759) -- but the lensmap build and the
 * per-frame lensmap apply run on an MI355X through libblinkyhip's C ABI (include/blinky_hip.h).
 * The six scene renders per frame stay in the engine's software rasteriser (R_RenderView).
 *
 * Host code stays C, as in the reference.  Every function names the reference lines it stands in for.
 */
#include "engine_iface.h"

#include <stdio.h>
#include <stdlib.h>
#include <errno.h>
#include <string.h>

#include "../../include/blinky_hip.h"

/* ---- globals the rest of the engine reads (fisheye.c:293-299) ---------------------------------- */
qboolean fisheye_enabled;
qboolean shortcutkeys_enabled;
double fisheye_plate_fov;

/* ---- state (fisheye.c:334-528, without the buffers: those live in HBM) --------------------------- */
/* The device side is ONE context, or - BLINKY_HIP_DEVICES="0,1,2,3" - a bk_multi that spreads the warp over several GPUs
 * in row stripes (each GPU builds and keeps its stripe of the lensmap, holds a replica of the globe, and copies its
 * rows of the frame straight into vid.buffer).  `bk` is the context scripts / plates are read back from: the single
 * context, or stripe context 0. */
static bk_ctx *bk;
static bk_multi *mg;
#define DEV(single, multi) (mg ? (multi) : (single))
static const char *dev_error(void) { return mg ? bk_multi_last_error(mg) : bk_last_error(bk); }
static int build_pending;                      /* bk_build answered BK_PENDING: the lens is compiling on another thread */
static struct { char name[50]; qboolean valid, changed; } globe, lens;
static struct { qboolean changed; int type, fov; } zoom;
static struct { qboolean enabled; int numcells; double cell_size, pad_size; } rubix;
static bk_plate plates[BK_MAX_PLATES];
static int numplates;
static int display[BK_MAX_PLATES];
static unsigned char palettes[BK_MAX_PLATES][256];
static int lensmap_ok;

#define VBUFFER(x, y) (vid.buffer + (x) + (y) * vid.rowbytes)      /* fisheye.c:634 */

static char *read_script(const char *kind, const char *name, size_t *len)
{
    char path[1024];
    FILE *f;
    long n;
    char *buf;
    snprintf(path, sizeof path, "%s/lua-scripts/%s/%s.lua", com_basedir, kind, name);   /* fisheye.c:1666, 1759 */
    f = fopen(path, "rb");
    if (!f) {
        /* what the reference prints when luaL_loadfile fails (fisheye.c:1671, 1764): Lua 5.2's LUA_ERRFILE and its message, no newline */
        Con_Printf("could not loadfile (%d) \nERROR: cannot open %s: %s", 7, path, strerror(errno));
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = (char *)malloc((size_t)n + 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
    fclose(f);
    buf[n] = 0;
    *len = (size_t)n;
    return buf;
}

/* what the scripts print()ed while loading goes where the reference's Lua VM sends it: stdout (lbaselib.c luaB_print) */
static void flush_script_output(void)
{
    static size_t shown;
    const char *text = bk ? bk_script_console(bk) : "";
    size_t n = strlen(text);
    if (n < shown) shown = 0;
    if (n > shown) {
        fputs(text + shown, stdout);
        fflush(stdout);
        shown = n;
    }
}

/* LUA_load_lens (fisheye.c:1659-1750) */
static qboolean load_lens(void)
{
    size_t len;
    char *src, chunk[80];
    int rc;
    if (!bk) return false;
    src = read_script("lenses", lens.name, &len);
    if (!src) { DEV(bk_clear_lens(bk), bk_multi_clear_lens(mg)); return false; }
    snprintf(chunk, sizeof chunk, "%s.lua", lens.name);
    rc = DEV(bk_load_lens(bk, src, len, chunk), bk_multi_load_lens(mg, src, len, chunk));
    free(src);
    flush_script_output();
    if (rc != BK_OK) { Con_Printf("%s\n", dev_error()); return false; }
    return true;
}

/* LUA_load_globe (fisheye.c:1752-1875) */
static qboolean load_globe(void)
{
    size_t len;
    char *src, chunk[80];
    int rc;
    numplates = 0;
    if (!bk) return false;
    src = read_script("globes", globe.name, &len);
    if (!src) { DEV(bk_clear_globe(bk), bk_multi_clear_globe(mg)); return false; }
    snprintf(chunk, sizeof chunk, "%s.lua", globe.name);
    rc = DEV(bk_load_globe(bk, src, len, chunk), bk_multi_load_globe(mg, src, len, chunk));
    free(src);
    flush_script_output();
    if (rc != BK_OK) { Con_Printf("%s\n", dev_error()); return false; }
    bk_get_globe(bk, plates, &numplates);
    return true;
}

/* ---- console commands (fisheye.c:916-1176) -------------------------------------------------------- */

/* The four zoom commands set one piece of state; the same table prints it ("Zoom currently: ...") and writes it to the config
 * (names, argument meaning and console text: fisheye.c:955-965, 1032-1058, 1273-1291, 683-696). */
static const struct zoom_cmd { int type; const char *name; const char *usage; } zoom_cmds[] = {
    {BK_ZOOM_FOV, "f_fov", "f_fov <degrees>: set horizontal FOV\n"},
    {BK_ZOOM_VFOV, "f_vfov", "f_vfov <degrees>: set vertical FOV\n"},
    {BK_ZOOM_COVER, "f_cover", NULL},             /* (no argument: no usage line) */
    {BK_ZOOM_CONTAIN, "f_contain", NULL},
};

/* the current zoom as the command that would set it ("f_vfov 100", "f_cover"), "" if none */
static void zoom_as_command(char *out, size_t n)
{
    size_t i;
    out[0] = 0;
    for (i = 0; i < sizeof zoom_cmds / sizeof zoom_cmds[0]; ++i) {
        if (zoom_cmds[i].type != zoom.type) continue;
        if (zoom_cmds[i].usage) snprintf(out, n, "%s %d", zoom_cmds[i].name, zoom.fov);
        else snprintf(out, n, "%s", zoom_cmds[i].name);
    }
}

static void zoom_command(const struct zoom_cmd *z)
{
    char now[32];
    if (z->usage && Cmd_Argc() < 2) {             /* a degrees command without degrees: say what it does and where the zoom stands */
        zoom_as_command(now, sizeof now);
        Con_Printf("%s", z->usage);
        Con_Printf("Zoom currently: %s\n", now[0] ? now : "none");
        return;
    }
    zoom.type = z->type;
    zoom.fov = z->usage ? (int)Q_atof(Cmd_Argv(1)) : 0;
    zoom.changed = true;
}
static void con_fov(void) { zoom_command(&zoom_cmds[0]); }
static void con_vfov(void) { zoom_command(&zoom_cmds[1]); }
static void con_cover(void) { zoom_command(&zoom_cmds[2]); }
static void con_contain(void) { zoom_command(&zoom_cmds[3]); }

static void con_dumppal(void)                       /* fisheye.c:916-931: the base palette as "r, g, b," lines in ./palette */
{
    int c;
    FILE *f = fopen("palette", "w");
    if (!f) { Con_Printf("could not open \"palette\" for writing\n"); return; }
    for (c = 0; c < 3 * 256; c += 3) fprintf(f, "%d, %d, %d,\n", host_basepal[c], host_basepal[c + 1], host_basepal[c + 2]);
    fclose(f);
}

static void con_rubix(void)                         /* fisheye.c:933-937 */
{
    rubix.enabled = !rubix.enabled;
    Con_Printf("Rubix is %s\n", rubix.enabled ? "ON" : "OFF");
}

static void con_rubixgrid(void)                     /* fisheye.c:939-953 */
{
    if (Cmd_Argc() != 4) {
        Con_Printf("RubixGrid <numcells> <cellsize> <padsize>\n"
                   "   numcells (default 10) = %d\n"
                   "   cellsize (default  4) = %f\n"
                   "   padsize  (default  1) = %f\n", rubix.numcells, rubix.cell_size, rubix.pad_size);
        return;
    }
    rubix.numcells = (int)Q_atof(Cmd_Argv(1));
    rubix.cell_size = Q_atof(Cmd_Argv(2));
    rubix.pad_size = Q_atof(Cmd_Argv(3));
    lens.changed = true;                            /* the grid is baked into the lensmap's tints: build again */
}

static void con_fisheye(void)                       /* fisheye.c:967-977 */
{
    if (Cmd_Argc() >= 2) {
        fisheye_enabled = Q_atoi(Cmd_Argv(1));       /* (the number as given: "fisheye 2" reads back, and is saved, as 2) */
        vid.recalc_refdef = true;
        return;
    }
    Con_Printf("Currently: fisheye %d\n\nTry F_HELP for more options and commands.\n", fisheye_enabled);
}

static void con_shortcutkeys(void)                  /* fisheye.c:979-1016 */
{
    static const char *lens_keys[9] = {"panini", "stereographic", "hammer", "winkeltripel", "fisheye1",
                                       "mercator", "quincuncial", "cube", "debug"};
    static const char *globe_keys[5][2] = {{"y", "cube"}, {"u", "cube_edge"}, {"i", "trism"}, {"o", "tetra"}, {"p", "fast"}};
    char cmd[96];
    int i;
    shortcutkeys_enabled = !shortcutkeys_enabled;
    if (shortcutkeys_enabled) {
        Con_Printf("Enabled Fisheye shortcut keys: 1-9 = Lenses, Y,U,I,O,P = Globes\n");
        for (i = 0; i < 9; ++i) {
            snprintf(cmd, sizeof cmd, "bind %d \"f_lens %s\"", i + 1, lens_keys[i]);
            Cmd_ExecuteString(cmd, src_command);
        }
        for (i = 0; i < 5; ++i) {
            snprintf(cmd, sizeof cmd, "bind %s \"f_globe %s\"", globe_keys[i][0], globe_keys[i][1]);
            Cmd_ExecuteString(cmd, src_command);
        }
    } else {
        Con_Printf("Disabled Fisheye shortcut keys\n");
        for (i = 1; i <= 8; ++i) {
            snprintf(cmd, sizeof cmd, "bind %d \"impulse %d\"", i, i);
            Cmd_ExecuteString(cmd, src_command);
        }
        Cmd_ExecuteString("unbind 9", src_command);
        for (i = 0; i < 5; ++i) {
            snprintf(cmd, sizeof cmd, "unbind %s", globe_keys[i][0]);
            Cmd_ExecuteString(cmd, src_command);
        }
    }
}

static void con_help(void)                          /* fisheye.c:1018-1030 */
{
    static const char *const lines[] = {
        "-----------------------------", "Welcome to the FISHEYE ADDON!", "-> fisheye 1    (ENABLE)", "-> fisheye 0    (DISABLE)", "",
        "-> f_lens <tab>    (CHANGE LENS)", "-> f_fov <degrees> (SET FOV)", "", "-> f_<tab>         (MORE COMMANDS)",
        "-----------------------------",
    };
    size_t i;
    for (i = 0; i < sizeof lines / sizeof lines[0]; ++i) Con_Printf("%s\n", lines[i]);
}

static void con_lens(void)                          /* fisheye.c:1061-1103 */
{
    bk_lens_info info;
    if (Cmd_Argc() < 2) {
        Con_Printf("f_lens <name>: use a new lens\n");
        Con_Printf("Currently: %s\n", lens.name);
        return;
    }
    lens.changed = true;
    snprintf(lens.name, sizeof lens.name, "%s", Cmd_Argv(1));
    Con_Printf("f_lens %s", lens.name);
    lens.valid = load_lens();
    if (!lens.valid) {
        strcpy(lens.name, "");
        Con_Printf("not a valid lens\n");
    }
    /* the lens' onload command string gives a friendly default view (e.g. "f_fov 180") */
    if (lens.valid && bk_get_lens_info(bk, &info) == BK_OK && info.onload[0]) {
        Cmd_ExecuteString(info.onload, src_command);
        Con_Printf("; %s\n", info.onload);
    } else {
        Con_Printf("\n");
    }
}

static struct { qboolean should; int with_margins; char name[32]; } save;       /* globe.save, fisheye.c:370-375 */

static void con_saveglobe(void)                     /* fisheye.c:1120-1136 */
{
    if (Cmd_Argc() < 2) {
        Con_Printf("f_saveglobe <name> [full flag=0]: screenshot the globe plates\n");
        return;
    }
    strncpy(save.name, Cmd_Argv(1), 32);
    save.name[31] = 0;
    save.with_margins = Cmd_Argc() >= 3 ? Q_atoi(Cmd_Argv(2)) : 0;
    save.should = true;
}

/* WritePCXplate, fisheye.c:1396-1465.  The plate image (texels, 0xFE outside the plate's own region unless
 * with_margins) comes from the device (bk_save_plate); header, run-length escapes and palette are packed here
 * exactly as the reference packs them (pcx_t, NQ/client.h:376-391; little-endian shorts). */
static void write_pcx_plate(const char *filename, int plate_index, int with_margins)
{
    int platesize = 0, width, height, i, j, length;
    byte *pcx, *pack, *img;
    const byte *palette = host_basepal;
    bk_get_size(bk, NULL, NULL, &platesize, NULL, NULL);
    width = height = platesize;
    pcx = (byte *)Hunk_TempAlloc(width * height * 2 + 1000);
    img = (byte *)malloc((size_t)width * height);
    if (pcx == NULL || img == NULL) {
        Con_Printf("SCR_ScreenShot_f: not enough memory\n");
        free(img);
        return;
    }
    if (bk_save_plate(bk, 0, plate_index, with_margins, img, width) != BK_OK) {
        Con_Printf("f_saveglobe: %s\n", bk_last_error(bk));
        free(img);
        return;
    }
    memset(pcx, 0, 128);
    pcx[0] = 0x0a;                                    /* manufacturer: PCX id */
    pcx[1] = 5;                                       /* version: 256 color */
    pcx[2] = 1;                                       /* encoding */
    pcx[3] = 8;                                       /* bits_per_pixel */
    pcx[8] = (byte)((width - 1) & 0xFF);   pcx[9] = (byte)((width - 1) >> 8);      /* xmax */
    pcx[10] = (byte)((height - 1) & 0xFF); pcx[11] = (byte)((height - 1) >> 8);    /* ymax */
    pcx[12] = (byte)(width & 0xFF);        pcx[13] = (byte)(width >> 8);           /* hres */
    pcx[14] = (byte)(height & 0xFF);       pcx[15] = (byte)(height >> 8);          /* vres */
    pcx[65] = 1;                                      /* color_planes: chunky image */
    pcx[66] = (byte)(width & 0xFF);        pcx[67] = (byte)(width >> 8);           /* bytes_per_line */
    pcx[68] = 2;                                      /* palette_type: not a grey scale */
    pack = pcx + 128;
    for (i = 0; i < height; i++)
        for (j = 0; j < width; j++) {
            byte col = img[(size_t)i * width + j];
            if ((col & 0xc0) == 0xc0) *pack++ = 0xc1;
            *pack++ = col;
        }
    *pack++ = 0x0c;                                   /* palette ID byte */
    for (i = 0; i < 768; i++) *pack++ = *palette++;
    length = (int)(pack - pcx);
    COM_WriteFile(filename, pcx, length);
    free(img);
}

static void save_globe(int numplates)                /* fisheye.c:1467-1484 */
{
    int i;
    char pcxname[32];
    save.should = false;
    D_EnableBackBufferAccess();
    for (i = 0; i < numplates; ++i) {
        snprintf(pcxname, 32, "%s%d.pcx", save.name, i);
        write_pcx_plate(pcxname, i, save.with_margins);
        Con_Printf("Wrote %s\n", pcxname);
    }
    D_DisableBackBufferAccess();
}

static void con_globe(void)                         /* fisheye.c:1138-1161 */
{
    if (Cmd_Argc() < 2) {
        Con_Printf("f_globe <name>: use a new globe\n");
        Con_Printf("Currently: %s\n", globe.name);
        return;
    }
    globe.changed = true;
    snprintf(globe.name, sizeof globe.name, "%s", Cmd_Argv(1));
    Con_Printf("f_globe %s\n", globe.name);
    globe.valid = load_globe();
    if (!globe.valid) {
        strcpy(globe.name, "");
        Con_Printf("not a valid globe\n");
    }
}

/* autocompletion for lens / globe names (fisheye.c:1105-1117, 1163-1175): the .lua files of the script directories */
static struct stree_root *cmdarg_scripts(const char *dir, const char *arg)
{
    struct stree_root *root = (struct stree_root *)Z_Malloc(sizeof(struct stree_root));
    if (root) {
        *root = STREE_ROOT;
        STree_AllocInit();
        COM_ScanDir(root, dir, arg, ".lua", true);
    }
    return root;
}
static struct stree_root *cmdarg_lens(const char *arg) { return cmdarg_scripts("../lua-scripts/lenses", arg); }
static struct stree_root *cmdarg_globe(const char *arg) { return cmdarg_scripts("../lua-scripts/globes", arg); }

/* ---- public functions (engine/include/fisheye.h) ----------------------------------------------------- */

void F_Init(void)                                   /* fisheye.c:642-676 */
{
    const char *dev = getenv("BLINKY_HIP_DEVICE"), *devs = getenv("BLINKY_HIP_DEVICES");
    rubix.enabled = false;
    /* compiled lens modules are cached under the USER's cache directory ($BLINKY_HIP_CACHE, else $XDG_CACHE_HOME/blinky_hip, else
     * ~/.cache/blinky_hip: the library's default) so that only the first run of a lens waits for hiprtc - never next to the game data:
     * com_basedir is where mods and downloaded content live, and the cache holds code the library loads */
    /* init_lua's counterpart: the script interpreter lives inside the context */
    if (devs && strchr(devs, ',')) {
        int list[16], n = 0;
        char buf[128], *tok;
        snprintf(buf, sizeof buf, "%s", devs);
        for (tok = strtok(buf, ","); tok && n < 16; tok = strtok(NULL, ",")) list[n++] = atoi(tok);
        mg = bk_create_multi(n, list);
        if (!mg) Con_Printf("fisheye: %s\n", bk_multi_last_error(NULL));
        else bk = bk_multi_ctx(mg, 0);
    } else {
        bk = bk_create(dev && !strcmp(dev, "none") ? BK_DEVICE_NONE : (dev ? atoi(dev) : (devs ? atoi(devs) : -1)));
        if (!bk) Con_Printf("fisheye: %s\n", bk_last_error(NULL));
    }
    /* BLINKY_HIP_RESIDENT=1: F_RenderView's per-frame calls - bk_upload_plate_async for every displayed plate, bk_apply - go through
     * the resident apply kernel (one kernel that stays on the GPU, a frame is a command): no apply launch per frame.
     * BLINKY_HIP_RESERVE_SLOTS=n keeps n workgroup places of every CU free - for other users of the GPU and for this context's own small
     * kernels (the plates' re-tiling, the frame's copy back), which then run beside the resident kernel as they do without it; the
     * default is 1.  With 0 the resident kernel may take every place: plates are re-tiled on the host and travel by DMA, the frame
     * comes back through a pinned copy (1 ms more host work per 4K frame, nothing per frame on the GPU but the warp). */
    if (bk && getenv("BLINKY_HIP_RESIDENT") && atoi(getenv("BLINKY_HIP_RESIDENT")) != 0) {
        const int reserve = getenv("BLINKY_HIP_RESERVE_SLOTS") ? atoi(getenv("BLINKY_HIP_RESERVE_SLOTS")) : 1;
        int i, rc = BK_OK;
        if (mg) {
            for (i = 0; i < bk_multi_size(mg) && rc == BK_OK; ++i) rc = bk_set_resident_share(bk_multi_ctx(mg, i), 0, 1, reserve);
            if (rc == BK_OK) rc = bk_multi_set_resident_apply(mg, 1);
        } else {
            rc = bk_set_resident_share(bk, 0, 1, reserve);
            if (rc == BK_OK) rc = bk_set_resident_apply(bk, 1);
        }
        if (rc != BK_OK) Con_Printf("fisheye: %s\n", dev_error());
    }
    /* the first use of a lens compiles it (hiprtc, 0.2-1.1 s): do that off the render thread and keep drawing with the
     * previous lensmap meanwhile - the reference's time-sliced builder never stalls a frame either (fisheye.c:2084-2217) */
    if (bk && !getenv("BLINKY_HIP_SYNC_COMPILE")) {
        int i;
        if (mg) for (i = 0; i < bk_multi_size(mg); ++i) bk_set_async_compile(bk_multi_ctx(mg, i), 1);
        else bk_set_async_compile(bk, 1);
    }

    Cmd_AddCommand("fisheye", con_fisheye);
    Cmd_AddCommand("f_help", con_help);
    Cmd_AddCommand("f_dumppal", con_dumppal);
    Cmd_AddCommand("f_rubix", con_rubix);
    Cmd_AddCommand("f_rubixgrid", con_rubixgrid);
    Cmd_AddCommand("f_cover", con_cover);
    Cmd_AddCommand("f_contain", con_contain);
    Cmd_AddCommand("f_fov", con_fov);
    Cmd_AddCommand("f_vfov", con_vfov);
    Cmd_AddCommand("f_lens", con_lens);
    Cmd_SetCompletion("f_lens", cmdarg_lens);                        /* fisheye.c:661 */
    Cmd_AddCommand("f_globe", con_globe);
    Cmd_SetCompletion("f_globe", cmdarg_globe);                      /* fisheye.c:663 */
    Cmd_AddCommand("f_saveglobe", con_saveglobe);
    Cmd_AddCommand("f_shortcutkeys", con_shortcutkeys);

    /* defaults */
    Cmd_ExecuteString("fisheye 1", src_command);
    Cmd_ExecuteString("f_globe cube", src_command);
    Cmd_ExecuteString("f_lens panini", src_command);
    Cmd_ExecuteString("f_fov 180", src_command);
    Cmd_ExecuteString("f_rubixgrid 10 4 1", src_command);

    if (host_basepal) bk_create_palmap(host_basepal, palettes);      /* create_palmap, fisheye.c:857-908 */
    if (!bk) fisheye_enabled = false;
}

void F_Shutdown(void)                               /* fisheye.c:678-681 */
{
    if (mg) bk_destroy_multi(mg); else bk_destroy(bk);
    mg = NULL;
    bk = NULL;
}

void F_WriteConfig(FILE *f)                         /* fisheye.c:683-696 */
{
    char z[32];
    zoom_as_command(z, sizeof z);
    fprintf(f, "fisheye %d\nf_lens \"%s\"\nf_globe \"%s\"\n", fisheye_enabled, lens.name, globe.name);
    fprintf(f, "f_rubixgrid %d %f %f\n", rubix.numcells, rubix.cell_size, rubix.pad_size);
    if (z[0]) fprintf(f, "%s\n", z);
}

/* render_plate (fisheye.c:2427-2450): the engine renders the view, its rows go to the GPU */
static void render_plate(int plate_index, vec3_t forward, vec3_t right, vec3_t up)
{
    VectorCopy(forward, r_refdef.forward);
    VectorCopy(right, r_refdef.right);
    VectorCopy(up, r_refdef.up);
    R_PushDlights();
    R_RenderView();
    /* pipelined: the rows go to a pinned buffer and the DMA runs while the engine renders the next plate */
    if (DEV(bk_upload_plate_async(bk, 0, plate_index, VBUFFER(scr_vrect.x, scr_vrect.y), vid.rowbytes),
            bk_multi_upload_plate(mg, 0, plate_index, VBUFFER(scr_vrect.x, scr_vrect.y), vid.rowbytes)) != BK_OK)
        Con_Printf("fisheye: %s\n", dev_error());
}

void F_RenderView(void)                             /* fisheye.c:698-811 */
{
    static int pwidth = -1, pheight = -1;
    int width_px = scr_vrect.width, height_px = scr_vrect.height;
    int sizechange = (pwidth != width_px) || (pheight != height_px);
    vec3_t forward, right, up;
    vrect_t vrect;
    int i;

    if (!bk) return;
    if (sizechange && DEV(bk_resize(bk, width_px, height_px), bk_multi_resize(mg, width_px, height_px)) != BK_OK) {   /* fisheye.c:712-727; no exit(1) */
        Con_Printf("Quake-Lenses: %s\n", dev_error());
        return;
    }
    if (sizechange || zoom.changed || lens.changed || globe.changed || build_pending) {      /* fisheye.c:730-743 */
        int rc, newdisplay[BK_MAX_PLATES];
        if (!build_pending || sizechange || zoom.changed || lens.changed || globe.changed) {
            /* the lens is loaded again so that variables depending on the globe (numplates) are fresh */
            /* (with no lens selected - the name is "" after an invalid one - the reference still tries, and says that ".lua" cannot be opened) */
            lens.valid = load_lens();
            if (!lens.valid) {
                strcpy(lens.name, "");
                Con_Printf("not a valid lens\n");
            }
            DEV(bk_set_zoom(bk, zoom.type, zoom.fov), bk_multi_set_zoom(mg, zoom.type, zoom.fov));
            DEV(bk_set_rubixgrid(bk, rubix.numcells, rubix.cell_size, rubix.pad_size),
                bk_multi_set_rubixgrid(mg, rubix.numcells, rubix.cell_size, rubix.pad_size));
        }
        for (i = 0; i < BK_MAX_PLATES; ++i) newdisplay[i] = 0;
        rc = DEV(bk_build(bk, newdisplay, NULL), bk_multi_build(mg, newdisplay, NULL));      /* create_lensmap, fisheye.c:2367-2397 */
        if (mg && rc == BK_OK) {
            /* several GPUs: cut the stripes by work, not by height (a lens that leaves part of the screen unmapped would
             * give the first and last GPU next to nothing), and build the lensmap of the new stripes.  Builds are
             * sub-millisecond; when the bounds do not move bk_set_rows keeps the maps and the second build is skipped. */
            int bounds[BK_MAX_PLATES * 8 + 1], n = bk_multi_size(mg);
            if (n > 1 && n < (int)(sizeof bounds / sizeof bounds[0]) && bk_multi_rebalance(mg, bounds) == BK_OK) {
                /* bk_set_rows invalidates the lensmap of every context whose stripe actually moved (also after a resize put
                 * the stripes back to equal shares): ask the contexts, not a remembered set of bounds */
                if (!bk_multi_lensmap_valid(mg)) rc = bk_multi_build(mg, newdisplay, NULL);
            }
        }
        build_pending = rc == BK_PENDING;
        if (!build_pending && !mg) {
            /* a lens whose callbacks do not become GPU code (recursion, run-time tables ...) or carry state is evaluated by the library's
             * interpreter on the host: say so once per build, with the construct that forced it */
            char why[512];
            const int path = bk_last_build_path(bk, why, sizeof why);
            if (path > 0) Con_Printf("lens callbacks evaluated on the host (%s): %s\n", path == 2 ? "sequential scan" : "worker pool", why);
        }
        if (!build_pending) {                    /* (while the new lens compiles the previous lensmap and its plates stay) */
            /* the reference clears the display flags only once calc_zoom has succeeded (fisheye.c:2376-2385): after a zoom it cannot
             * compute, or with no valid lens / globe, the plates of the previous lensmap go on being rendered (into a lensmap that shows
             * none of them) - kept, so that the engine sees the same R_RenderView calls */
            /* a lens_inverse that returned a malformed result ends the reference's scan there: what it had set by then stays on screen
             * and the display flags are those of that partial table (fisheye.c:2113-2115, 1976) - the library reports BK_E_SCRIPT with
             * exactly that table in place and its flags in newdisplay */
            const int partial = rc == BK_E_SCRIPT && (mg ? bk_multi_lensmap_valid(mg) : bk_last_build_bad_key(bk) != 0);
            if (rc == BK_OK || partial) for (i = 0; i < numplates; ++i) display[i] = newdisplay[i];     /* (only the current globe's plates are reset) */
            lensmap_ok = rc == BK_OK || partial;
            if (rc != BK_OK && lens.valid && globe.valid) Con_Printf("%s\n", dev_error());
        }
    }

    AngleVectors(r_refdef.viewangles, forward, right, up);                  /* fisheye.c:750 */
    vrect.x = 0;
    vrect.y = 0;
    vrect.width = vid.width;
    vrect.height = vid.height;
    vrect.pnext = NULL;
    R_SetVrect(&vrect, &scr_vrect, sb_lines);

    for (i = 0; i < numplates; ++i) {                                        /* fisheye.c:764-794 */
        if (display[i]) {
            vec3_t r = {0, 0, 0}, u = {0, 0, 0}, f = {0, 0, 0};
            fisheye_plate_fov = plates[i].fov;
            R_ViewChanged(&vrect, sb_lines, vid.aspect);
            VectorMA(r, plates[i].right[0], right, r);
            VectorMA(r, plates[i].right[1], up, r);
            VectorMA(r, plates[i].right[2], forward, r);
            VectorMA(u, plates[i].up[0], right, u);
            VectorMA(u, plates[i].up[1], up, u);
            VectorMA(u, plates[i].up[2], forward, u);
            VectorMA(f, plates[i].forward[0], right, f);
            VectorMA(f, plates[i].forward[1], up, f);
            VectorMA(f, plates[i].forward[2], forward, f);
            render_plate(i, f, r, u);
        }
    }

    if (save.should) save_globe(numplates);                                  /* fisheye.c:797-799 */

    Draw_TileClear(0, 0, vid.width, vid.height);                             /* fisheye.c:802 */
    /* render_lensmap, fisheye.c:2406-2424 */
    if (DEV(bk_apply(bk, 0, vid.buffer, vid.rowbytes, scr_vrect.x, scr_vrect.y, rubix.enabled, palettes),
            bk_multi_apply(mg, 0, vid.buffer, vid.rowbytes, scr_vrect.x, scr_vrect.y, rubix.enabled, palettes)) != BK_OK && lensmap_ok)
        Con_Printf("fisheye: %s\n", dev_error());

    pwidth = width_px;
    pheight = height_px;
    lens.changed = globe.changed = zoom.changed = false;                    /* fisheye.c:810 */
}
