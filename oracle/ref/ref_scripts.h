/* ref_scripts.h -- ORACLE test infrastructure: script provider for luashim.c */
#ifndef REF_SCRIPTS_H
#define REF_SCRIPTS_H
#include "fakelua/lua.h"
int  ref_script_exists(const char *path);
void ref_script_run(lua_State *L, const char *path);   /* "executes the chunk": sets globals */
#endif
