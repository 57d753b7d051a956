/*
 * engine_iface.h -- the TyrQuake services fisheye_hip.c uses.
 *
 * Inside the engine tree build with -DBLINKY_IN_ENGINE: the real headers are included and this
 * file adds nothing.  Stand-alone (this repository's tests) it declares the same symbols with
 * the reference's signatures, each citing where the engine declares it (paths relative to
 * /root/reference/engine); tests/host/engine_stub.c provides stand-ins.
 */
#ifndef BLINKY_ENGINE_IFACE_H
#define BLINKY_ENGINE_IFACE_H

#ifdef BLINKY_IN_ENGINE
#include "cmd.h"
#include "common.h"
#include "console.h"
#include "draw.h"
#include "host.h"
#include "mathlib.h"
#include "quakedef.h"
#include "r_local.h"
#include "screen.h"
#include "sbar.h"
#include "vid.h"
#else

#include <stdio.h>

typedef enum { false, true } qboolean;          /* include/qtypes.h */
typedef unsigned char byte;
typedef float vec_t;                            /* include/mathlib.h:30-31 */
typedef vec_t vec3_t[3];
typedef byte pixel_t;                           /* include/vid.h:31 */

typedef struct vrect_s {                        /* include/vid.h:33-36 */
    int x, y, width, height;
    struct vrect_s *pnext;
} vrect_t;

typedef struct {                                /* include/vid.h:38-57 (fields used here) */
    pixel_t *buffer;
    int rowbytes;
    int width, height;
    float aspect;
    int recalc_refdef;
} viddef_t;

typedef struct {                                /* include/render.h:125-141 (fields used here; forward/right/up */
    vec3_t viewangles;                          /*  are the fisheye patch's additions, fisheye.patch)             */
    vec3_t forward, right, up;
} refdef_t;

typedef void (*xcommand_t)(void);               /* include/cmd.h:76 */
typedef enum { src_client, src_command } cmd_source_t;   /* include/cmd.h:90-95 */
struct rb_root { struct rb_node *rb_node; };              /* include/rb_tree.h:50-54 */
#define QRB_ROOT (struct rb_root) { NULL, }
struct stree_root {                                       /* include/shell.h:50-58 */
    unsigned int entries;
    unsigned int maxlen;
    unsigned int minlen;
    struct rb_root root;
    struct stree_stack *stack;
};
#define STREE_ROOT (struct stree_root) { 0, 0, -1, QRB_ROOT, NULL }
typedef struct stree_root *(*cmd_arg_f)(const char *);   /* include/cmd.h:84 */

extern viddef_t vid;                            /* include/vid.h:59 */
extern refdef_t r_refdef;                       /* include/render.h:152 */
extern vrect_t scr_vrect;                       /* include/screen.h:47 */
extern int sb_lines;                            /* include/sbar.h:29 */
extern byte *host_basepal;                      /* NQ/host.h:40 */
extern char com_basedir[];                      /* include/common.h:201 */

void Cmd_AddCommand(const char *cmd_name, xcommand_t function);      /* include/cmd.h:115 */
void Cmd_SetCompletion(const char *cmd_name, cmd_arg_f completion);  /* include/cmd.h:116 */
void Cmd_ExecuteString(const char *text, cmd_source_t src);          /* include/cmd.h:103 */
int Cmd_Argc(void);                                                  /* include/cmd.h:131 */
const char *Cmd_Argv(int arg);                                       /* include/cmd.h:132 */
void Con_Printf(const char *fmt, ...);                               /* include/console.h:51 */
float Q_atof(const char *str);                                       /* include/common.h:165 */
int Q_atoi(const char *str);                                         /* include/common.h:164 */
void AngleVectors(const vec3_t angles, vec3_t forward, vec3_t right, vec3_t up);   /* include/mathlib.h */
void VectorMA(const vec3_t veca, const float scale, const vec3_t vecb, vec3_t vecc);   /* common/mathlib.c:350 */
void R_PushDlights(void);                                            /* include/render.h:191 */
void R_RenderView(void);                                             /* include/render.h:162 */
void R_ViewChanged(vrect_t *pvrect, int lineadj, float aspect);      /* include/render.h:163 */
void R_SetVrect(const vrect_t *pvrectin, vrect_t *pvrect, int lineadj);   /* include/render.h:211 */
void Draw_TileClear(int x, int y, int w, int h);                     /* include/draw.h:42 */
void *Hunk_TempAlloc(int size);                                      /* include/zone.h:112 */
void *Z_Malloc(int size);                                            /* include/zone.h:96 (zero-filled) */
void STree_AllocInit(void);                                          /* include/shell.h:68 */
void COM_ScanDir(struct stree_root *root, const char *path, const char *pfx, const char *ext, qboolean stripext);   /* include/common.h:206 */
void COM_WriteFile(const char *filename, const void *data, int len); /* include/common.h:204 */
void D_EnableBackBufferAccess(void);                                 /* include/d_iface.h:140 */
void D_DisableBackBufferAccess(void);

#define VectorCopy(a, b) do { (b)[0] = (a)[0]; (b)[1] = (a)[1]; (b)[2] = (a)[2]; } while (0)   /* mathlib.h:73 */
#endif /* BLINKY_IN_ENGINE */

#endif
