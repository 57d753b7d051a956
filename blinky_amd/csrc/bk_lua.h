// bk_lua.h -- from-scratch front-end for the Lua 5.2 subset that Blinky lens / globe scripts
// use: lexer, parser -> AST with resolved locals/upvalues, and a host
// tree-walking interpreter.  The reference links the stock Lua 5.2 VM (engine/Makefile:818);
// that dependency is absent here, and the GPU needs the callbacks as compiled code anyway,
// so one AST feeds two back-ends: this interpreter (chunk execution, calc_zoom, globe
// loading: fisheye.c:1659-1875, 1293-1386) and the HIP emitter in bk_emit.cpp (per-pixel
// callbacks: fisheye.c:1545-1651).
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace bklua {

struct LuaError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- math backend -----------------------------------------------------------------------
// Scripts see math.sin etc.  The product uses the portable bkm.h functions (bit-identical to
// the device build); a platform-libm table exists so that test infrastructure can run the
// interpreter the way the stock Lua VM would (math.sin == libm sin).
struct MathLib {
    double (*sin)(double), (*cos)(double), (*tan)(double);
    double (*asin)(double), (*acos)(double), (*atan)(double), (*atan2)(double, double);
    double (*sinh)(double), (*cosh)(double), (*tanh)(double);
    double (*exp)(double), (*log)(double), (*log10)(double), (*pow)(double, double);
    double (*sqrt)(double), (*fmod)(double, double);
};
const MathLib &math_portable();   // bkm.h
const MathLib &math_platform();
const MathLib &math_perturbed(double rel, int mode);      // test-only, see bk_lua.cpp   // <cmath>

// ---- AST -----------------------------------------------------------------------------------
struct Expr;
struct Stmt;
struct FuncProto;
using ExprP = std::unique_ptr<Expr>;
using StmtP = std::unique_ptr<Stmt>;
using Block = std::vector<StmtP>;

enum class VarKind { Local, Upvalue, Global };

struct Expr {
    enum Kind { Nil, True, False, Number, String, Vararg, Name, Index, Call, Binop, Unop, Function, Table } kind;
    int line = 0;
    double num = 0;
    std::string str;              // String value / Name / operator
    // the operator of a Binop / Unop once more as a number: the interpreter dispatches on it (string compares per
    // evaluated operator were most of its time)
    enum Op { OP_NONE, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_POW, OP_CONCAT, OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_AND, OP_OR,
              OP_NOT, OP_NEG, OP_LEN, OP_PAREN };
    Op op = OP_NONE;
    // Name
    VarKind var = VarKind::Global;
    int slot = -1;                // local slot or upvalue index; Global: the name's process-wide number (global_id)
    // Index: a[b]; Call: a(args); Binop: a op b; Unop: op a
    ExprP a, b;
    std::vector<ExprP> args;      // Call arguments / Table array items
    std::vector<std::pair<ExprP, ExprP>> fields;   // Table [k]=v / name=v entries, in source order
    std::vector<int> item_order;  // Table: >=0 -> args[i] (positional), <0 -> fields[-1-i]
    FuncProto *proto = nullptr;   // Function
};

struct Stmt {
    enum Kind { Local, Assign, CallStmt, Do, While, Repeat, If, NumFor, GenFor, Return, Break, LocalFunction, Goto, Label } kind;   // (Goto / Label: names[0])
    int line = 0;
    std::vector<int> slots;               // Local / NumFor(var) / GenFor(vars) / LocalFunction
    std::vector<std::string> names;
    std::vector<ExprP> targets;           // Assign
    std::vector<ExprP> exprs;             // Local / Assign / Return / NumFor(start,stop[,step]) / GenFor explist
    ExprP call;                           // CallStmt
    Block body;                           // Do / While / Repeat / NumFor / GenFor
    ExprP cond;                           // While / Repeat
    std::vector<std::pair<ExprP, Block>> clauses;   // If: (cond, block); else has null cond
};

struct UpvalDesc { bool from_parent_local; int index; std::string name; };

struct FuncProto {
    std::string name;                     // for messages / emitted symbol
    int line = 0;
    int id = 0;
    int nparams = 0;
    bool is_vararg = false;
    int nslots = 0;                       // local slots (params first)
    std::vector<std::string> slot_names;
    std::vector<UpvalDesc> upvals;
    std::vector<char> captured;           // by local slot: an inner function refers to it (it then lives in a shared cell)
    bool is_captured(int slot) const { return (size_t)slot < captured.size() && captured[(size_t)slot]; }
    Block body;
    FuncProto *parent = nullptr;
};

struct Chunk {
    std::string name;
    std::vector<std::unique_ptr<FuncProto>> protos;   // protos[0] = main
    FuncProto *main() const { return protos[0].get(); }
};

std::shared_ptr<Chunk> parse(const std::string &src, const std::string &chunkname);
int global_id(const std::string &name);     // process-wide number of a global name (assigned by the parser)

// ---- runtime values --------------------------------------------------------------------------
struct Table;
struct Closure;
struct Interp;
struct Builtin;

// A Lua value: 32 bytes - tag, number / boolean, and ONE reference for the heap kinds (string, table, closure, builtin).
// Numbers and booleans - nearly everything a lens callback touches - copy without touching a reference count.
struct Value {
    enum T : uint8_t { NIL, BOOL, NUM, STR, TABLE, FUNC, BUILTIN, THREAD } t = NIL;
    bool b = false;
    double n = 0;
    std::shared_ptr<void> p;                // STR: std::string, TABLE: Table, FUNC: Closure, BUILTIN: Builtin, THREAD: Coroutine (bk_lua.cpp)

    static Value nil() { return Value(); }
    static Value boolean(bool v) { Value x; x.t = BOOL; x.b = v; return x; }
    static Value number(double v) { Value x; x.t = NUM; x.n = v; return x; }
    static Value string(const std::string &v) { Value x; x.t = STR; x.p = std::make_shared<std::string>(v); return x; }
    static Value table(std::shared_ptr<Table> tp) { Value x; x.t = TABLE; x.p = std::move(tp); return x; }
    static Value closure(std::shared_ptr<Closure> c) { Value x; x.t = FUNC; x.p = std::move(c); return x; }
    static Value builtin(std::shared_ptr<Builtin> bp) { Value x; x.t = BUILTIN; x.p = std::move(bp); return x; }
    const std::string &str() const { return *static_cast<const std::string *>(p.get()); }
    Table *tab() const { return static_cast<Table *>(p.get()); }
    Closure *fn() const { return static_cast<Closure *>(p.get()); }
    Builtin *bi() const { return static_cast<Builtin *>(p.get()); }
    std::shared_ptr<Table> tab_ptr() const { return std::static_pointer_cast<Table>(p); }
    std::shared_ptr<Closure> fn_ptr() const { return std::static_pointer_cast<Closure>(p); }
    std::shared_ptr<Builtin> bi_ptr() const { return std::static_pointer_cast<Builtin>(p); }
    bool truthy() const { return !(t == NIL || (t == BOOL && !b)); }
    bool is_function() const { return t == FUNC || t == BUILTIN; }
    const char *type_name() const;
};

// Argument / result lists: a vector with room for a few values inside the object - a callback evaluation makes
// dozens of calls, and a heap allocation per argument list, result list and frame was most of its time.
template <size_t kInline>
class ValuesN {
    using Values = ValuesN;
public:
    ValuesN() {}
    ValuesN(std::initializer_list<Value> il) { reserve(il.size()); for (const Value &v : il) push_back(v); }
    ValuesN(const Values &o) { reserve(o.n_); for (size_t i = 0; i < o.n_; ++i) new (data_ + i) Value(o.data_[i]); n_ = o.n_; }
    ValuesN(Values &&o) noexcept { steal(o); }
    Values &operator=(const Values &o) { if (this != &o) { clear(); reserve(o.n_); for (size_t i = 0; i < o.n_; ++i) new (data_ + i) Value(o.data_[i]); n_ = o.n_; } return *this; }
    Values &operator=(Values &&o) noexcept { if (this != &o) { clear(); release(); steal(o); } return *this; }
    ~ValuesN() { clear(); release(); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    Value &operator[](size_t i) { return data_[i]; }
    const Value &operator[](size_t i) const { return data_[i]; }
    Value &back() { return data_[n_ - 1]; }
    const Value &back() const { return data_[n_ - 1]; }
    Value *begin() { return data_; }
    Value *end() { return data_ + n_; }
    const Value *begin() const { return data_; }
    const Value *end() const { return data_ + n_; }
    void clear() { for (size_t i = 0; i < n_; ++i) data_[i].~Value(); n_ = 0; }
    void reserve(size_t want)
    {
        if (want <= cap_) return;
        size_t cap = cap_ * 2 > want ? cap_ * 2 : want;
        Value *nd = static_cast<Value *>(::operator new(cap * sizeof(Value)));
        for (size_t i = 0; i < n_; ++i) { new (nd + i) Value(std::move(data_[i])); data_[i].~Value(); }
        release();
        data_ = nd;
        cap_ = cap;
    }
    void push_back(const Value &v) { if (n_ == cap_) { Value tmp(v); reserve(n_ + 1); new (data_ + n_++) Value(std::move(tmp)); return; } new (data_ + n_++) Value(v); }
    void push_back(Value &&v) { if (n_ == cap_) { Value tmp(std::move(v)); reserve(n_ + 1); new (data_ + n_++) Value(std::move(tmp)); return; } new (data_ + n_++) Value(std::move(v)); }
    void resize(size_t n)
    {
        reserve(n);
        while (n_ > n) data_[--n_].~Value();
        while (n_ < n) new (data_ + n_++) Value();
    }
    void append(const Value *first, const Value *last) { reserve(n_ + (size_t)(last - first)); for (; first != last; ++first) new (data_ + n_++) Value(*first); }
    void assign(const Value *first, const Value *last) { clear(); append(first, last); }

private:
    bool inlined() const { return data_ == reinterpret_cast<const Value *>(buf_); }
    void release() { if (!inlined()) ::operator delete(data_); data_ = reinterpret_cast<Value *>(buf_); cap_ = kInline; }
    void steal(Values &o)
    {
        if (o.inlined()) {
            data_ = reinterpret_cast<Value *>(buf_); cap_ = kInline;
            for (size_t i = 0; i < o.n_; ++i) { new (data_ + i) Value(std::move(o.data_[i])); o.data_[i].~Value(); }
        } else { data_ = o.data_; cap_ = o.cap_; o.data_ = reinterpret_cast<Value *>(o.buf_); o.cap_ = kInline; }
        n_ = o.n_;
        o.n_ = 0;
    }
    alignas(Value) unsigned char buf_[kInline * sizeof(Value)];
    Value *data_ = reinterpret_cast<Value *>(buf_);
    size_t n_ = 0, cap_ = kInline;
};
using Values = ValuesN<4>;

using BuiltinFn = std::function<void(Interp &, const Values &args, Values &rets)>;
// a builtin; `f1` set: a one-argument math function (math.sin ...) the interpreter calls without building argument lists
struct Builtin { std::string name; BuiltinFn fn; double (*MathLib::*f1)(double) = nullptr; };

struct Table {
    std::vector<Value> arr;                 // t[1..n]
    std::map<double, Value> nhash;          // other numeric keys
    std::map<std::string, Value> shash;
    std::shared_ptr<Table> meta;            // setmetatable: consulted where a plain table would give nil or an error (host interpreter only)
    Value get(const Value &k) const;
    void set(const Value &k, const Value &v);
    size_t length() const { return arr.size(); }
};

struct Closure {
    FuncProto *proto = nullptr;
    std::shared_ptr<Chunk> chunk;           // keeps the AST alive
    std::vector<std::shared_ptr<Value>> upvals;
};

struct Interp {
    explicit Interp(const MathLib &m);
    struct Empty {};
    Interp(const MathLib &m, Empty) : math(&m) {}        // no standard library: clone() fills the globals itself
    const MathLib *math;                    // swappable: platform libm (reference-faithful) or bkm.h
    std::map<std::string, Value> globals;   // (a nil-valued entry is an absent global; nodes are never erased, see gslots)
    std::vector<Value *> gslots;            // by Expr::slot of a Global name: its node in `globals` (std::map nodes do not move)
    Value &global_ref(int gid, const std::string &name)
    {
        if ((size_t)gid >= gslots.size()) gslots.resize((size_t)gid + 64, nullptr);
        Value *&p = gslots[(size_t)gid];
        if (!p) p = &globals[name];
        return *p;
    }
    long steps = 0, max_steps = 200000000;  // runaway-script guard
    unsigned long long activity = 0;        // bumped whenever script code runs or the host changes a global: "nothing a script can see has changed since"
    std::string goto_label;                 // the label a `goto` under way is looking for
    const std::string *call_chunk = nullptr; // where the builtin call being made stands (error() puts "chunk:line:" in front of its message)
    int call_line = 0;
    int depth = 0;
    Value registry;                         // debug.getregistry's table, made when first asked for (a copy of the interpreter starts with none)
    void *current_co = nullptr;             // the coroutine whose body is running on this interpreter (coroutine library, bk_lua.cpp); null = the main thread
    std::function<void(const std::string &)> print_sink;   // `print` / io.write output, newlines included (Con_Printf)

    Value get_global(const std::string &name) const;
    void set_global(const std::string &name, const Value &v);
    void register_builtin(const std::string &name, BuiltinFn fn);
    // run a chunk (luaL_loadbuffer + lua_pcall)
    void run(const std::string &src, const std::string &chunkname);
    // call any function value (lua_call with LUA_MULTRET)
    Values call(const Value &f, const Values &args);
    std::string tostring(const Value &v) const;
    // An independent copy of this interpreter's state for another thread: globals, tables, closures and their
    // upvalue cells are deep-copied (ASTs, strings and builtins are immutable and stay shared); `roots` are values
    // of this interpreter (e.g. the lens callbacks) whose counterparts in the copy are returned in roots_out.
    // print() in the copy is silent.
    std::unique_ptr<Interp> clone(const Values &roots, Values *roots_out) const;
};

}  // namespace bklua
