/* fakelua/lualib.h -- ORACLE test infrastructure (see fakelua/lua.h). */
#ifndef FAKELUA_LUALIB_H
#define FAKELUA_LUALIB_H
#include "lua.h"
void luaL_openlibs(lua_State *L);
#endif
