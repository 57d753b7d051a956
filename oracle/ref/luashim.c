/*
 * luashim.c -- ORACLE test infrastructure.
 * A just-big-enough implementation of the Lua 5.2 C API subset declared in
 * fakelua/lua.h: a value stack with call frames, a globals table, a registry,
 * and array-only tables.  It contains no script semantics; "loading a file"
 * asks the script provider (ref_script_exists / ref_script_run) which fills in
 * globals through this same API.
 */
#include "fakelua/lua.h"
#include "fakelua/lauxlib.h"
#include "fakelua/lualib.h"
#include "ref_scripts.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { V_NIL, V_NUM, V_STR, V_CFUNC, V_TABLE, V_CHUNK };

typedef struct Table Table;
typedef struct {
    int t;
    double n;
    const char *s;          /* strings are interned copies, never freed (test tool) */
    lua_CFunction f;
    Table *tab;
} Value;
struct Table { int n, cap; Value *arr; };

#define STACK_MAX 1024
#define GLOBALS_MAX 256

struct lua_State {
    Value stack[STACK_MAX];
    int top, base;          /* absolute indices; frame-relative index 1 == stack[base] */
    struct { const char *name; Value v; } globals[GLOBALS_MAX];
    int nglobals;
    Value *registry;        /* grows: the reference never gives a reference back (no luaL_unref in fisheye.c), and a harness
                             * process may load thousands of lenses */
    int nreg, regcap;
};

static void die(const char *msg) { fprintf(stderr, "luashim: %s\n", msg); abort(); }

static Value *at(lua_State *L, int idx)
{
    static Value nil_value;
    int a = idx > 0 ? L->base + idx - 1 : L->top + idx;
    nil_value.t = V_NIL;
    if (a < L->base || a >= L->top) return &nil_value;   /* acceptable index with no value */
    return &L->stack[a];
}
static void push(lua_State *L, Value v)
{
    if (L->top >= STACK_MAX) die("stack overflow");
    L->stack[L->top++] = v;
}
static Value mk(int t) { Value v; memset(&v, 0, sizeof v); v.t = t; return v; }

lua_State *luaL_newstate(void) { return (lua_State *)calloc(1, sizeof(lua_State)); }
void luaL_openlibs(lua_State *L) { (void)L; }
void lua_close(lua_State *L) { free(L); }

int lua_gettop(lua_State *L) { return L->top - L->base; }
void lua_settop(lua_State *L, int idx)
{
    int newtop = idx >= 0 ? L->base + idx : L->top + idx + 1;
    if (newtop < L->base) die("settop below frame");
    while (L->top < newtop) push(L, mk(V_NIL));
    L->top = newtop;
}
void lua_pushnil(lua_State *L) { push(L, mk(V_NIL)); }
void lua_pushnumber(lua_State *L, lua_Number n) { Value v = mk(V_NUM); v.n = n; push(L, v); }
void lua_pushinteger(lua_State *L, lua_Integer n) { lua_pushnumber(L, (lua_Number)n); }
void lua_pushstring(lua_State *L, const char *s) { Value v = mk(V_STR); v.s = strdup(s); push(L, v); }
void lua_pushcclosure(lua_State *L, lua_CFunction f, int n) { Value v = mk(V_CFUNC); (void)n; v.f = f; push(L, v); }

void lua_createtable(lua_State *L, int narr, int nrec)
{
    Value v = mk(V_TABLE);
    (void)nrec;
    v.tab = (Table *)calloc(1, sizeof(Table));
    v.tab->cap = narr > 4 ? narr : 4;
    v.tab->arr = (Value *)calloc((size_t)v.tab->cap, sizeof(Value));
    push(L, v);
}
void lua_rawseti(lua_State *L, int idx, int n)   /* t[n] = pop() */
{
    Value *t = at(L, idx);
    if (t->t != V_TABLE || n < 1) die("rawseti on non-table");
    if (n > t->tab->cap) {
        int nc = n * 2;
        t->tab->arr = (Value *)realloc(t->tab->arr, (size_t)nc * sizeof(Value));
        memset(t->tab->arr + t->tab->cap, 0, (size_t)(nc - t->tab->cap) * sizeof(Value));
        t->tab->cap = nc;
    }
    t->tab->arr[n - 1] = L->stack[L->top - 1];
    if (n > t->tab->n) t->tab->n = n;
    L->top--;
}
void lua_rawgeti(lua_State *L, int idx, int n)
{
    if (idx == LUA_REGISTRYINDEX) {
        if (n < 0 || n >= L->nreg) { lua_pushnil(L); return; }
        push(L, L->registry[n]);
        return;
    }
    {
        Value *t = at(L, idx);
        if (t->t != V_TABLE) die("rawgeti on non-table");
        if (n < 1 || n > t->tab->n) lua_pushnil(L); else push(L, t->tab->arr[n - 1]);
    }
}
size_t lua_rawlen(lua_State *L, int idx)
{
    Value *v = at(L, idx);
    if (v->t == V_TABLE) return (size_t)v->tab->n;
    if (v->t == V_STR) return strlen(v->s);
    return 0;
}
/* array-part iteration in index order, as Lua 5.2 does for a pure sequence */
int lua_next(lua_State *L, int idx)
{
    Value *t = at(L, idx);
    Value key = L->stack[L->top - 1];
    int k;
    if (t->t != V_TABLE) die("next on non-table");
    L->top--;
    k = key.t == V_NIL ? 1 : (int)key.n + 1;
    if (k > t->tab->n) return 0;
    lua_pushnumber(L, k);
    push(L, t->tab->arr[k - 1]);
    return 1;
}
int lua_type(lua_State *L, int idx)
{
    switch (at(L, idx)->t) {
    case V_NUM: return LUA_TNUMBER;
    case V_STR: return LUA_TSTRING;
    case V_CFUNC: case V_CHUNK: return LUA_TFUNCTION;
    case V_TABLE: return LUA_TTABLE;
    default: return LUA_TNIL;
    }
}
int lua_isnumber(lua_State *L, int idx) { return at(L, idx)->t == V_NUM; }
int lua_isstring(lua_State *L, int idx) { int t = at(L, idx)->t; return t == V_STR || t == V_NUM; }
lua_Number lua_tonumberx(lua_State *L, int idx, int *isnum)
{
    Value *v = at(L, idx);
    if (isnum) *isnum = v->t == V_NUM;
    return v->t == V_NUM ? v->n : 0;
}
lua_Integer lua_tointegerx(lua_State *L, int idx, int *isnum) { return (lua_Integer)lua_tonumberx(L, idx, isnum); }
const char *lua_tolstring(lua_State *L, int idx, size_t *len)
{
    Value *v = at(L, idx);
    if (v->t != V_STR) return NULL;
    if (len) *len = strlen(v->s);
    return v->s;
}
lua_Number luaL_checknumber(lua_State *L, int arg)
{
    if (at(L, arg)->t != V_NUM) die("luaL_checknumber: not a number");
    return at(L, arg)->n;
}
void lua_getglobal(lua_State *L, const char *name)
{
    int i;
    for (i = 0; i < L->nglobals; ++i)
        if (!strcmp(L->globals[i].name, name)) { push(L, L->globals[i].v); return; }
    lua_pushnil(L);
}
void lua_setglobal(lua_State *L, const char *name)
{
    int i;
    Value v = L->stack[--L->top];
    for (i = 0; i < L->nglobals; ++i)
        if (!strcmp(L->globals[i].name, name)) { L->globals[i].v = v; return; }
    if (L->nglobals >= GLOBALS_MAX) die("too many globals");
    L->globals[L->nglobals].name = strdup(name);
    L->globals[L->nglobals++].v = v;
}
int luaL_ref(lua_State *L, int t)
{
    if (t != LUA_REGISTRYINDEX) die("luaL_ref: registry only");
    if (L->nreg == 0) L->nreg = 1;         /* ref 0 unused, like real Lua */
    if (L->nreg >= L->regcap) {
        L->regcap = L->regcap ? 2 * L->regcap : 256;
        L->registry = (Value *)realloc(L->registry, (size_t)L->regcap * sizeof(Value));
        if (!L->registry) die("luaL_ref: out of memory");
    }
    L->registry[L->nreg] = L->stack[--L->top];
    return L->nreg++;
}
int luaL_loadbufferx(lua_State *L, const char *buff, size_t sz, const char *name, const char *mode)
{
    Value v = mk(V_CHUNK);
    (void)buff; (void)sz; (void)mode;
    v.s = strdup(name);      /* the alias chunk (fisheye.c:1230-1248) needs no action here */
    push(L, v);
    return 0;
}
int luaL_loadfilex(lua_State *L, const char *filename, const char *mode)
{
    (void)mode;
    if (!ref_script_exists(filename)) {
        char msg[512];
        snprintf(msg, sizeof msg, "cannot open %s: No such file or directory", filename);   /* Lua 5.2 lauxlib.c errfile(): "cannot %s %s: %s" */
        lua_pushstring(L, msg);
        return 7;            /* LUA_ERRFILE */
    }
    {
        Value v = mk(V_CHUNK);
        v.s = strdup(filename);
        push(L, v);
    }
    return 0;
}
/* call: stack = ... f a1..an  ->  ... r1..rm */
void lua_callk(lua_State *L, int nargs, int nresults, int ctx, lua_CFunction k)
{
    int fpos = L->top - nargs - 1;
    Value f = L->stack[fpos];
    int saved_base = L->base, nret, i, first;
    (void)ctx; (void)k;
    if (f.t == V_CHUNK) {
        L->top = fpos;
        ref_script_run(L, f.s);
        if (nresults > 0) for (i = 0; i < nresults; ++i) lua_pushnil(L);
        return;
    }
    if (f.t != V_CFUNC) die("call of a non-function");
    L->base = fpos + 1;
    nret = f.f(L);
    first = L->top - nret;
    for (i = 0; i < nret; ++i) L->stack[fpos + i] = L->stack[first + i];
    L->top = fpos + nret;
    L->base = saved_base;
    if (nresults != LUA_MULTRET) {
        while (nret < nresults) { lua_pushnil(L); nret++; }
        if (nret > nresults) L->top -= nret - nresults;
    }
}
int lua_pcallk(lua_State *L, int nargs, int nresults, int errfunc, int ctx, lua_CFunction k)
{
    (void)errfunc;
    lua_callk(L, nargs, nresults, ctx, k);
    return 0;
}
