/*
 * ref_scripts.c -- ORACLE test infrastructure: script provider for oracle/_ref.
 *
 * "Running" <basedir>/lua-scripts/{lenses,globes}/<name>.lua under the fake Lua
 * state means: publish the globals that script would leave behind, taken from
 * the hand transliterations in ../oracle_lenses.c.  The lens functions become
 * lua_CFunctions; when they need latlon_to_ray / ray_to_latlon / plate_to_ray
 * they call the globals the REFERENCE registered (fisheye.c:1257-1264), so the
 * float rounding at the C<->Lua boundary is the reference's own code.
 */
#include "ref_scripts.h"
#include "../oracle.h"

#include <stdio.h>
#include <string.h>

static lua_State *cur_L;
static ok_lens_def cur_lens;
static ok_host ref_host;

/* ---- ok_host -> the reference's registered C functions ------------------------ */
static void h_latlon_to_ray(void *ctx, double lat, double lon, double out[3])
{
    lua_State *L = (lua_State *)ctx;
    lua_getglobal(L, "latlon_to_ray");
    lua_pushnumber(L, lat);
    lua_pushnumber(L, lon);
    lua_call(L, 2, 3);
    out[0] = lua_tonumber(L, -3); out[1] = lua_tonumber(L, -2); out[2] = lua_tonumber(L, -1);
    lua_pop(L, 3);
}
static void h_ray_to_latlon(void *ctx, double x, double y, double z, double *lat, double *lon)
{
    lua_State *L = (lua_State *)ctx;
    lua_getglobal(L, "ray_to_latlon");
    lua_pushnumber(L, x);
    lua_pushnumber(L, y);
    lua_pushnumber(L, z);
    lua_call(L, 3, 2);
    *lat = lua_tonumber(L, -2); *lon = lua_tonumber(L, -1);
    lua_pop(L, 2);
}
static int h_plate_to_ray(void *ctx, double plate, double u, double v, double out[3])
{
    lua_State *L = (lua_State *)ctx;
    int top = lua_gettop(L), n;
    lua_getglobal(L, "plate_to_ray");
    lua_pushnumber(L, plate);
    lua_pushnumber(L, u);
    lua_pushnumber(L, v);
    lua_call(L, 3, LUA_MULTRET);
    n = lua_gettop(L) - top;
    if (n == 3) { out[0] = lua_tonumber(L, -3); out[1] = lua_tonumber(L, -2); out[2] = lua_tonumber(L, -1); }
    lua_pop(L, n);
    return n == 3;
}

/* ---- lens callbacks as lua_CFunctions ----------------------------------------- */
static int L_lens_inverse(lua_State *L)
{
    double ray[3];
    int st = cur_lens.inverse(&ref_host, lua_tonumber(L, 1), lua_tonumber(L, 2), ray);
    if (st == 1) { lua_pushnumber(L, ray[0]); lua_pushnumber(L, ray[1]); lua_pushnumber(L, ray[2]); return 3; }
    lua_pushnil(L);
    return 1;
}
static int L_lens_forward(lua_State *L)
{
    double x, y;
    int st = cur_lens.forward(&ref_host, lua_tonumber(L, 1), lua_tonumber(L, 2), lua_tonumber(L, 3), &x, &y);
    if (st == 1) { lua_pushnumber(L, x); lua_pushnumber(L, y); return 2; }
    lua_pushnil(L);
    return 1;
}

static ok_globe_plate_fn cur_globe_plate;
static int L_globe_plate(lua_State *L)
{
    int plate = 0;
    if (cur_globe_plate(&ref_host, lua_tonumber(L, 1), lua_tonumber(L, 2), lua_tonumber(L, 3), &plate)) { lua_pushnumber(L, plate); return 1; }
    lua_pushnil(L);
    return 1;
}

/* ---- path -> (kind, name) ------------------------------------------------------ */
static int split(const char *path, char *kind, char *name)
{
    const char *p = strstr(path, "/lua-scripts/");
    const char *slash, *dot;
    if (!p) return 0;
    p += strlen("/lua-scripts/");
    slash = strchr(p, '/');
    if (!slash) return 0;
    dot = strrchr(slash, '.');
    if (!dot || strcmp(dot, ".lua")) return 0;
    snprintf(kind, 16, "%.*s", (int)(slash - p), p);
    snprintf(name, 64, "%.*s", (int)(dot - slash - 1), slash + 1);
    return 1;
}

int ref_script_exists(const char *path)
{
    char kind[16], name[64];
    ok_lens_def d;
    ok_globe_def g;
    if (!split(path, kind, name)) return 0;
    if (!strcmp(kind, "lenses")) return ok_find_lens(name, &d);
    if (!strcmp(kind, "globes")) return ok_find_globe(name, &g);
    return 0;
}

static void setnum(lua_State *L, const char *k, double v) { lua_pushnumber(L, v); lua_setglobal(L, k); }

void ref_script_run(lua_State *L, const char *path)
{
    char kind[16], name[64];
    if (!split(path, kind, name)) return;        /* e.g. the "aliases" chunk */
    cur_L = L;
    ref_host.latlon_to_ray = h_latlon_to_ray;
    ref_host.ray_to_latlon = h_ray_to_latlon;
    ref_host.plate_to_ray = h_plate_to_ray;
    ref_host.ctx = L;
    if (!strcmp(kind, "lenses")) {
        int np = 0;
        lua_getglobal(L, "numplates");                       /* published by LUA_load_lens before the chunk runs */
        np = (int)lua_tonumber(L, -1);
        lua_pop(L, 1);
        ok_set_script_env(&ref_host, np);
        if (!ok_find_lens(name, &cur_lens)) return;
        if (cur_lens.max_fov) setnum(L, "max_fov", cur_lens.max_fov);
        if (cur_lens.max_vfov) setnum(L, "max_vfov", cur_lens.max_vfov);
        if (cur_lens.width != 0) setnum(L, "lens_width", cur_lens.width);
        if (cur_lens.height != 0) setnum(L, "lens_height", cur_lens.height);
        if (cur_lens.onload) { lua_pushstring(L, cur_lens.onload); lua_setglobal(L, "onload"); }
        if (cur_lens.inverse) { lua_pushcfunction(L, L_lens_inverse); lua_setglobal(L, "lens_inverse"); }
        if (cur_lens.forward) { lua_pushcfunction(L, L_lens_forward); lua_setglobal(L, "lens_forward"); }
    } else if (!strcmp(kind, "globes")) {
        ok_globe_def g;
        int i, j;
        if (!ok_find_globe(name, &g)) return;
        if (!strcmp(name, "tetra")) {                        /* tetra.lua:19 print(fovd): luaB_print -> "%.14g" and a newline on stdout, flushed */
            printf("%.14g\n", g.fov_deg[0]);
            fflush(stdout);
        }
        lua_createtable(L, g.numplates, 0);                  /* plates = { {fwd,up,fov}, ... } */
        for (i = 0; i < g.numplates; ++i) {
            lua_createtable(L, 3, 0);
            lua_createtable(L, 3, 0);
            for (j = 0; j < 3; ++j) { lua_pushnumber(L, g.forward[i][j]); lua_rawseti(L, -2, j + 1); }
            lua_rawseti(L, -2, 1);
            lua_createtable(L, 3, 0);
            for (j = 0; j < 3; ++j) { lua_pushnumber(L, g.up[i][j]); lua_rawseti(L, -2, j + 1); }
            lua_rawseti(L, -2, 2);
            lua_pushnumber(L, g.fov_deg[i]);
            lua_rawseti(L, -2, 3);
            lua_rawseti(L, -2, i + 1);
        }
        lua_setglobal(L, "plates");
        cur_globe_plate = g.globe_plate;
        if (g.globe_plate) { lua_pushcfunction(L, L_globe_plate); lua_setglobal(L, "globe_plate"); }
    }
}
