/* fakelua/lauxlib.h -- ORACLE test infrastructure (see fakelua/lua.h). */
#ifndef FAKELUA_LAUXLIB_H
#define FAKELUA_LAUXLIB_H
#include "lua.h"
lua_State *luaL_newstate(void);
int  luaL_loadbufferx(lua_State *L, const char *buff, size_t sz, const char *name, const char *mode);
int  luaL_loadfilex(lua_State *L, const char *filename, const char *mode);
int  luaL_ref(lua_State *L, int t);
lua_Number luaL_checknumber(lua_State *L, int arg);
#define luaL_loadbuffer(L,s,sz,n) luaL_loadbufferx(L, s, sz, n, NULL)
#define luaL_loadfile(L,f)        luaL_loadfilex(L, f, NULL)
#endif
