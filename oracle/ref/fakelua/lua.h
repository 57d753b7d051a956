/*
 * fakelua/lua.h -- ORACLE test infrastructure.
 * Lua 5.2 is an un-vendored dependency of the reference (engine/Makefile:818,
 * BUILDING.md) and is absent from this image.  This header declares just the
 * part of the Lua 5.2 C API that engine/NQ/fisheye.c uses, so that the
 * UNMODIFIED fisheye.c compiles; luashim.c implements it (value stack, globals,
 * registry, array tables).  Script chunks are supplied by ref_scripts.c.
 */
#ifndef FAKELUA_LUA_H
#define FAKELUA_LUA_H
#include <stddef.h>

typedef struct lua_State lua_State;
typedef double lua_Number;
typedef ptrdiff_t lua_Integer;
typedef int (*lua_CFunction)(lua_State *L);

#define LUA_MULTRET (-1)
#define LUA_REGISTRYINDEX (-1001000)

#define LUA_TNIL 0
#define LUA_TBOOLEAN 1
#define LUA_TNUMBER 3
#define LUA_TSTRING 4
#define LUA_TTABLE 5
#define LUA_TFUNCTION 6

int  lua_gettop(lua_State *L);
void lua_settop(lua_State *L, int idx);
void lua_pushnil(lua_State *L);
void lua_pushnumber(lua_State *L, lua_Number n);
void lua_pushinteger(lua_State *L, lua_Integer n);
void lua_pushstring(lua_State *L, const char *s);
void lua_pushcclosure(lua_State *L, lua_CFunction f, int n);
void lua_rawgeti(lua_State *L, int idx, int n);
void lua_rawseti(lua_State *L, int idx, int n);
void lua_createtable(lua_State *L, int narr, int nrec);
size_t lua_rawlen(lua_State *L, int idx);
int  lua_next(lua_State *L, int idx);
int  lua_type(lua_State *L, int idx);
int  lua_isnumber(lua_State *L, int idx);
int  lua_isstring(lua_State *L, int idx);
lua_Number lua_tonumberx(lua_State *L, int idx, int *isnum);
lua_Integer lua_tointegerx(lua_State *L, int idx, int *isnum);
const char *lua_tolstring(lua_State *L, int idx, size_t *len);
void lua_getglobal(lua_State *L, const char *name);
void lua_setglobal(lua_State *L, const char *name);
void lua_callk(lua_State *L, int nargs, int nresults, int ctx, lua_CFunction k);
int  lua_pcallk(lua_State *L, int nargs, int nresults, int errfunc, int ctx, lua_CFunction k);
void lua_close(lua_State *L);

#define lua_call(L,n,r)        lua_callk(L, (n), (r), 0, NULL)
#define lua_pcall(L,n,r,f)     lua_pcallk(L, (n), (r), (f), 0, NULL)
#define lua_pop(L,n)           lua_settop(L, -(n)-1)
#define lua_tonumber(L,i)      lua_tonumberx(L, (i), NULL)
#define lua_tointeger(L,i)     lua_tointegerx(L, (i), NULL)
#define lua_tostring(L,i)      lua_tolstring(L, (i), NULL)
#define lua_pushcfunction(L,f) lua_pushcclosure(L, (f), 0)
#define lua_isfunction(L,n)    (lua_type(L, (n)) == LUA_TFUNCTION)
#define lua_istable(L,n)       (lua_type(L, (n)) == LUA_TTABLE)
#define lua_isnil(L,n)         (lua_type(L, (n)) == LUA_TNIL)
#endif
