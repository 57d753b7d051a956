// bk_emit.cpp -- Lua AST -> HIP C++.
//
// The per-pixel callbacks of a lens / globe script (lens_inverse, lens_forward, globe_plate:
// fisheye.c:1545-1651) and every script function they reach are translated into device
// functions over tagged doubles (bk_device_rt.h).  The translation is a direct, statement by
// statement restatement of the interpreter in bk_lua.cpp - same operation order, every
// arithmetic operation a single IEEE double operation or a bkm.h call - so the device result
// is bit-identical to what the host interpreter computes for the same arguments.
//
// Script globals: a global that device code never assigns is a constant (its value after the
// chunk ran and calc_zoom called lens_forward); one that device code assigns becomes a
// per-thread variable initialised from that value (the shipped scripts only use such globals
// as scratch within one call or as pure caches).
#include "bk_emit.h"

#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <sstream>

namespace bk {
using namespace bklua;

namespace {

[[noreturn]] void unsupported(const std::string &chunk, int line, const std::string &what)
{
    throw LuaError(chunk + ":" + std::to_string(line) + ": not supported in a GPU callback: " + what);
}

std::string sanitize(const std::string &s)
{
    std::string o;
    for (char c : s) o += (isalnum((unsigned char)c) ? c : '_');
    return o;
}

std::string num_literal(double d)
{
    if (d != d) return "bk_num(__builtin_nan(\"\"))";
    if (std::isinf(d)) return d > 0 ? "bk_num(__builtin_inf())" : "bk_num(-__builtin_inf())";
    char buf[64];
    snprintf(buf, sizeof buf, "bk_num(%a)", d);
    return buf;
}

struct FnInfo {
    const FuncProto *proto = nullptr;
    const Closure *cl = nullptr;
    std::string cname;
    bool emitting = false, done = false;
};

struct Emitter {
    Interp &I;
    std::map<std::pair<const FuncProto *, std::string>, FnInfo> fns;   // (a function once per set of function-valued arguments it is called with)
    std::vector<std::string> fn_code;                 // in dependency order
    std::set<std::string> mutable_globals;
    std::map<const Table *, std::pair<std::string, int>> const_tables;   // table -> (array name, n)
    std::ostringstream table_code;
    int uid = 0;
    // sin(v) and cos(v) of the same operand share an argument reduction (bk_f_sincos: the values are the two single calls' bit for
    // bit) - inside a statement and from one plain assignment to the next (`local c = cos(phi)  local s = sin(phi)`).  An entry holds
    // for the C++ scope and the stretch of straight-line code it was made in: `epoch` moves on at every block, compound statement,
    // short-circuit branch and call of a script function (which may assign the operand), and an assignment to the operand forgets it.
    struct SinCos { std::string s, c; long epoch; };
    std::map<std::string, SinCos> sincos;
    long epoch = 0;
    // (r6) The same memory for every other pure operation - arithmetic, comparisons, the one-argument math functions, atan2 - keyed by
    // the call as it is written out ("bk_mul(S, l4, sn15)"): `1/tan(lat)` twice in two statements, or `sin(lon*sin(lat))` next to
    // `cos(lon*sin(lat))` (polyconic.lua), are evaluated once.  Each is a function of its operands' values alone (the flags it may raise
    // are raised the first time), so the rules are the sin / cos ones: same epoch, no operand assigned since.  Not while the
    // derivative of a loop body is being written out (every temporary there carries its own derivative note).
    struct Pure { std::string name; long epoch; };
    std::map<std::string, Pure> pure;
    std::string last_single, last_single_of;          // emit_call: the value of a one-result builtin call (when it is a plain name), and the result array it was the call of
    static bool is_name(const std::string &v)
    {
        if (v.empty() || (v[0] >= '0' && v[0] <= '9')) return false;
        for (char c : v) if (!(c == '_' || c == '.' || (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'))) return false;
        return true;
    }
    static bool mentions(const std::string &key, const std::string &var)
    {
        auto word = [](char c) { return c == '_' || c == '.' || (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
        for (size_t at = key.find(var); at != std::string::npos; at = key.find(var, at + 1)) {
            const bool left = at == 0 || !word(key[at - 1]), right = at + var.size() >= key.size() || !word(key[at + var.size()]);
            if (left && right) return true;
        }
        return false;
    }
    void forget(const std::string &var)
    {
        sincos.erase(var);
        for (auto it = pure.begin(); it != pure.end();) it = mentions(it->first, var) ? pure.erase(it) : std::next(it);
    }
    // string constants: device code only ever compares them, so a string is its number in this table
    std::map<std::string, int> strings{{"nil", 1}, {"boolean", 2}, {"number", 3}, {"string", 4}};   // (type() results first)
    std::string str_literal(const std::string &v)
    {
        auto it = strings.find(v);
        if (it == strings.end()) it = strings.emplace(v, (int)strings.size() + 1).first;
        return "bk_str(" + std::to_string(it->second) + ")";
    }

    explicit Emitter(Interp &i) : I(i) {}

    // ---- per-function state --------------------------------------------------------------
    // a function being walked: a script function known when the code is generated (cl: its closure, upvalues are the chunk's
    // cells) or a function defined inside one (cl null: its upvalues are locals of `parent`, or what `parent` sees)
    struct Scope {
        const FuncProto *proto = nullptr;
        const Closure *cl = nullptr;
        const Scope *parent = nullptr;
    };
    struct Fn : Scope {
        std::ostringstream out;
        int indent = 1;
        std::map<int, int> array_slots;               // local slot -> capacity (array tables)
        std::map<int, std::string> fn_slots;          // local slot -> the lambda a `local function` / `local f = function` became
        std::set<int> fn_open;                        // ... whose body is being emitted right now (a call from inside is recursion)
        std::map<int, Value> static_slots;            // local slot -> the script function / builtin it holds for its whole life
                                                      // (a parameter bound at the call, `local f = math.sin`): calls resolve when the code is generated
        std::map<int, std::vector<std::string>> record_slots;   // local slot -> field names of `local p = {x = .., y = ..}`: one variable per field
        std::map<int, std::pair<int, int>> matrix_slots;         // local slot -> (rows, columns) of `local m = {{..}, {..}}`: one flat array
        std::string lp = "l", ap = "A", rp = "R";     // names of locals / local arrays / record fields (functions defined inside others get their own)
        std::string field_var(int slot, const std::string &name) const { return rp + std::to_string(slot) + "_" + name; }
        bool is_table(int slot) const { return array_slots.count(slot) || record_slots.count(slot) || matrix_slots.count(slot); }
        std::string chunk;
    };
    // what upvalue `idx` of `s` is: a local of an enclosing function being emitted (*owner, *slot), or (returned) a chunk cell
    static const Value *upvalue_of(const Scope *s, int idx, const Scope **owner, int *slot)
    {
        while (!s->cl) {
            const UpvalDesc &d = s->proto->upvals[(size_t)idx];
            if (d.from_parent_local) { *owner = s->parent; *slot = d.index; return nullptr; }
            idx = d.index;
            s = s->parent;
        }
        *owner = nullptr;
        return s->cl->upvals[(size_t)idx].get();
    }
    // is `e` the name of a local variable of f or of a function around it?  -> that function, the slot
    // Self-correcting iterations (`for i = 1, 20 do dt = -f(t) / f'(t); t = t + dt end`).  The bound of bk_device_rt.h treats t and
    // dt as independent and grows by |d dt / d t| per step - thousands where f' is small (eckert4's rows near the poles: 640 000
    // flagged pixels at 4K) - while a Newton step FORGETS most of the error it starts with.  For a loop whose body is straight-line
    // arithmetic with ONE variable carried from step to step, the body is differentiated along with being evaluated (plain doubles,
    // forward mode): the carried variable enters a step as exact, every variable the step assigns then gets
    //   bound = 2 |d variable / d carried| * (the carried variable's bound at the start of the step) + (its bound within the step)
    // (bk_contract).  First order, like every other bound here; the test is tests/test_exactness_cpu.py's adversarial libm.
    bool ad_active = false;
    Fn *ad_fn = nullptr;
    std::map<int, std::string> ad_slot;           // local slot -> its derivative variable
    std::map<std::string, std::string> ad_d;      // value expression (a temp, a local) -> derivative expression
    std::string ad_call;                          // derivative of the builtin call emit_call has just emitted ("" = none)
    std::string d_of(const std::string &v) const { auto it = ad_d.find(v); return it == ad_d.end() ? std::string("0.0") : it->second; }
    void ad_note(Fn &f, const std::string &value, const std::string &dexpr)
    {
        const std::string dn = tmp("dd");
        line(f, "const double " + dn + " = " + dexpr + ";");
        ad_d[value] = dn;
    }
    static Fn *local_of(Fn &f, const Expr &e, int *slot)
    {
        if (e.kind != Expr::Name) return nullptr;
        if (e.var == VarKind::Local) { *slot = e.slot; return &f; }
        if (e.var == VarKind::Upvalue) {
            const Scope *owner = nullptr;
            if (!upvalue_of(&f, e.slot, &owner, slot)) return const_cast<Fn *>(static_cast<const Fn *>(owner));
        }
        return nullptr;
    }
    // chunk cells that device code assigns: per-thread state like the assigned globals (cell -> field of BkState)
    std::vector<std::pair<const Value *, std::string>> mutable_cells;
    const std::string *cell_field(const Value *cell) const
    {
        for (auto &c : mutable_cells) if (c.first == cell) return &c.second;
        return nullptr;
    }
    std::string tmp(const char *p = "t") { return std::string(p) + std::to_string(++uid); }
    static void line(Fn &f, const std::string &s) { f.out << std::string((size_t)f.indent * 4, ' ') << s << "\n"; }
    std::string pure_value(Fn &f, const std::string &call)
    {
        auto hit = pure.find(call);
        if (hit != pure.end() && hit->second.epoch == epoch) return hit->second.name;
        const std::string t = tmp();
        line(f, "bkv " + t + " = " + call + ";");
        pure.insert_or_assign(call, Pure{t, epoch});
        return t;
    }

    // t[key] of a table known now, as the interpreter finds it: the table's own field, else along metatables whose __index is a table
    static Value static_field(const Value &t, const Value &key)
    {
        Value cur = t;
        for (int depth = 0; depth < 32 && cur.t == Value::TABLE; ++depth) {
            Value v = cur.tab()->get(key);
            if (v.t != Value::NIL || !cur.tab()->meta) return v;
            cur = cur.tab()->meta->get(Value::string("__index"));
        }
        return Value();
    }

    // ---- static resolution of a callee / constant ------------------------------------------
    // value a Name evaluates to at build time; `known` false for locals
    bool static_value(Fn &f, const Expr &e, Value *v)
    {
        if (e.kind == Expr::Name) {
            int lslot = 0;
            if (Fn *o = local_of(f, e, &lslot)) {
                auto it = o->static_slots.find(lslot);
                if (it == o->static_slots.end()) return false;
                *v = it->second;
                return true;
            }
            if (e.var == VarKind::Global) {
                if (mutable_globals.count(e.str)) return false;
                *v = I.get_global(e.str);
                return true;
            }
            if (e.var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                int slot = 0;
                const Value *cell = upvalue_of(&f, e.slot, &owner, &slot);
                if (!cell || cell_field(cell)) return false;
                *v = *cell;
                return true;
            }
            return false;
        }
        if (e.kind == Expr::Index && e.b->kind == Expr::String) {
            Value o;
            if (static_value(f, *e.a, &o) && o.t == Value::TABLE) { *v = static_field(o, Value::string(e.b->str)); return true; }
        }
        return false;
    }

    // is local `slot` of `p` ever the target of an assignment (a captured one: taken to be)?
    static bool assigned_in(const Block &b, int slot)
    {
        for (const StmtP &sp : b) {
            const Stmt &s = *sp;
            for (auto &t : s.targets) if (t->kind == Expr::Name && t->var == VarKind::Local && t->slot == slot) return true;
            if (assigned_in(s.body, slot)) return true;
            for (auto &c : s.clauses) if (assigned_in(c.second, slot)) return true;
        }
        return false;
    }
    static bool slot_is_constant(const FuncProto *p, int slot) { return !p->is_captured(slot) && !assigned_in(p->body, slot); }

    // ---- pre-pass: which globals does device code assign? ------------------------------------
    void scan_closure(const Closure *cl, std::set<const FuncProto *> &seen)
    {
        Scope sc;
        sc.proto = cl->proto;
        sc.cl = cl;
        scan_block(cl->proto->body, sc, seen);
    }
    void scan_block(const Block &b, const Scope &sc, std::set<const FuncProto *> &seen)
    {
        for (const StmtP &s : b) scan_stmt(*s, sc, seen);
    }
    void scan_expr(const Expr *e, const Scope &sc, std::set<const FuncProto *> &seen)
    {
        if (!e) return;
        if (e->kind == Expr::Call && e->str.empty()) {
            Value callee;
            bool known = false;
            if (e->a->kind == Expr::Name && e->a->var == VarKind::Global) { callee = I.get_global(e->a->str); known = true; }
            else if (e->a->kind == Expr::Name && e->a->var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                int slot = 0;
                if (const Value *cell = upvalue_of(&sc, e->a->slot, &owner, &slot)) { callee = *cell; known = true; }
            }
            if (known && callee.t == Value::FUNC && !seen.count(callee.fn()->proto)) {
                seen.insert(callee.fn()->proto);
                scan_closure(callee.fn(), seen);
            }
        }
        if (e->kind == Expr::Call && !e->str.empty() && e->a->kind == Expr::Name) {          // obj:m(..): the method's body is device code too
            Value obj;
            if (e->a->var == VarKind::Global) obj = I.get_global(e->a->str);
            else if (e->a->var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                int slot = 0;
                if (const Value *cell = upvalue_of(&sc, e->a->slot, &owner, &slot)) obj = *cell;
            }
            const Value m = static_field(obj, Value::string(e->str));
            if (m.t == Value::FUNC && !seen.count(m.fn()->proto)) {
                seen.insert(m.fn()->proto);
                scan_closure(m.fn(), seen);
            }
        }
        if (e->kind == Expr::Function && e->proto) {                  // a function defined inside device code: its body is device code
            Scope inner;
            inner.proto = e->proto;
            inner.parent = &sc;
            scan_block(e->proto->body, inner, seen);
        }
        scan_expr(e->a.get(), sc, seen);
        scan_expr(e->b.get(), sc, seen);
        for (auto &x : e->args) scan_expr(x.get(), sc, seen);
        for (auto &x : e->fields) { scan_expr(x.first.get(), sc, seen); scan_expr(x.second.get(), sc, seen); }
    }
    void scan_stmt(const Stmt &s, const Scope &sc, std::set<const FuncProto *> &seen)
    {
        for (auto &t : s.targets) {
            if (t->kind == Expr::Name && t->var == VarKind::Global) mutable_globals.insert(t->str);
            if (t->kind == Expr::Name && t->var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                int slot = 0;
                const Value *cell = upvalue_of(&sc, t->slot, &owner, &slot);
                if (cell && !cell_field(cell)) mutable_cells.emplace_back(cell, "u" + std::to_string(mutable_cells.size() + 1) + "_" + sanitize(t->str));
            }
            scan_expr(t.get(), sc, seen);
        }
        for (auto &x : s.exprs) scan_expr(x.get(), sc, seen);
        scan_expr(s.call.get(), sc, seen);
        scan_expr(s.cond.get(), sc, seen);
        scan_block(s.body, sc, seen);
        for (auto &c : s.clauses) { scan_expr(c.first.get(), sc, seen); scan_block(c.second, sc, seen); }
    }

    // ---- constant (never assigned) global / upvalue tables -----------------------------------
    std::pair<std::string, int> const_table(Fn &f, const Expr &at, const std::shared_ptr<Table> &t, const std::string &hint)
    {
        auto it = const_tables.find(t.get());
        if (it != const_tables.end()) return it->second;
        if (!t->nhash.empty() || !t->shash.empty())
            unsupported(f.chunk, at.line, "table '" + hint + "' with non-array keys");
        std::string name = "GT" + std::to_string(++uid) + "_" + sanitize(hint);
        int n = (int)t->arr.size();
        table_code << "static __device__ const bkv " << name << "[" << n + 1 << "] = {{0.0, 0.0, BK_TNIL}";
        for (const Value &v : t->arr) {
            char buf[80];
            switch (v.t) {
            case Value::NUM:
                if (v.n != v.n || std::isinf(v.n)) unsupported(f.chunk, at.line, "non-finite number in table '" + hint + "'");
                snprintf(buf, sizeof buf, ", {%a, 0.0, BK_TNUM}", v.n);
                break;
            case Value::BOOL: snprintf(buf, sizeof buf, ", {0.0, 0.0, %s}", v.b ? "BK_TTRUE" : "BK_TFALSE"); break;
            case Value::NIL: snprintf(buf, sizeof buf, ", {0.0, 0.0, BK_TNIL}"); break;
            case Value::STR: {
                const std::string lit = str_literal(v.str());        // "bk_str(<id>)"
                snprintf(buf, sizeof buf, ", {%s.0, 0.0, BK_TSTR}", lit.substr(7, lit.size() - 8).c_str());
                break;
            }
            default: unsupported(f.chunk, at.line, std::string("table '") + hint + "' holding a " + v.type_name());
            }
            table_code << buf;
        }
        table_code << "};\n";
        return const_tables[t.get()] = {name, n};
    }

    // ---- expressions ----------------------------------------------------------------------------
    std::string const_value(Fn &f, const Expr &e, const Value &v, const std::string &what)
    {
        switch (v.t) {
        case Value::NIL: return "bk_nil()";
        case Value::BOOL: return v.b ? "bk_bool(true)" : "bk_bool(false)";
        case Value::NUM: return num_literal(v.n);
        case Value::STR: return str_literal(v.str());
        default: unsupported(f.chunk, e.line, what + " (a " + v.type_name() + ") used as a value");
        }
    }

    std::string emit_expr(Fn &f, const Expr &e)
    {
        switch (e.kind) {
        case Expr::Nil: return "bk_nil()";
        case Expr::True: return "bk_bool(true)";
        case Expr::False: return "bk_bool(false)";
        case Expr::Number: return num_literal(e.num);
        case Expr::String: return str_literal(e.str);       // (compared with == / ~= only; no string operations on the device)
        case Expr::Vararg: {                           // in a single-value position: the first extra argument, or nil
            const std::string np = std::to_string(f.proto->nparams);
            return "(na > " + np + " ? a[" + np + "] : bk_nil())";
        }
        case Expr::Function: unsupported(f.chunk, e.line, "function values / closures");
        case Expr::Table: unsupported(f.chunk, e.line, "table constructors other than 'local t = {a, b, ...}'");
        case Expr::Name: {
            int slot = 0;
            if (Fn *o = local_of(f, e, &slot)) {
                if (o->is_table(slot)) unsupported(f.chunk, e.line, "table '" + e.str + "' used as a value");
                if (o->fn_slots.count(slot)) unsupported(f.chunk, e.line, "function '" + e.str + "' used as a value (a function defined inside a callback can only be called)");
                if (o->static_slots.count(slot))
                    unsupported(f.chunk, e.line, std::string(o->static_slots[slot].t == Value::TABLE ? "table" : "function") + " '" + e.str +
                                                     "' used as a value (it can be " + (o->static_slots[slot].t == Value::TABLE ? "indexed" : "called") + ", and passed on to script functions)");
                return o->lp + std::to_string(slot);
            }
            if (e.var == VarKind::Global && mutable_globals.count(e.str)) return "S.g_" + sanitize(e.str);
            if (e.var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                if (const std::string *field = cell_field(upvalue_of(&f, e.slot, &owner, &slot))) return "S." + *field;
            }
            {
                Value v;
                static_value(f, e, &v);
                return const_value(f, e, v, "'" + e.str + "'");
            }
        }
        case Expr::Index: {
            int aslot = 0;
            Fn *ao = local_of(f, *e.a, &aslot);
            if (ao && ao->record_slots.count(aslot)) {                         // p.x: the field's variable; a field the constructor did not name is nil
                if (e.b->kind != Expr::String) unsupported(f.chunk, e.line, "indexing record '" + e.a->str + "' with a computed key");
                const auto &names = ao->record_slots[aslot];
                for (const std::string &n : names) if (n == e.b->str) return ao->field_var(aslot, n);
                return "bk_nil()";
            }
            if (ao && ao->matrix_slots.count(aslot)) unsupported(f.chunk, e.line, "a row of table '" + e.a->str + "' used as a value");
            if (e.a->kind == Expr::Index) {                                    // m[i][j] of a local {{..}, {..}}
                int mslot = 0;
                Fn *mo = local_of(f, *e.a->a, &mslot);
                if (mo && mo->matrix_slots.count(mslot)) {
                    const std::string row = matrix_row(f, *mo, mslot, *e.a->b);
                    const std::string k = emit_expr(f, *e.b), t = tmp();
                    line(f, "bkv " + t + " = bk_aget(S, " + row + ", " + std::to_string(mo->matrix_slots[mslot].second) + ", " + k + ");");
                    return t;
                }
            }
            if (ao && ao->array_slots.count(aslot)) {
                std::string k = emit_expr(f, *e.b), t = tmp();
                line(f, "bkv " + t + " = bk_aget(S, " + ao->ap + std::to_string(aslot) + ", " + std::to_string(ao->array_slots[aslot]) + ", " + k + ");");
                return t;
            }
            Value sv;
            if (static_value(f, e, &sv)) return const_value(f, e, sv, "field '" + e.b->str + "'");
            Value o;
            if (static_value(f, *e.a, &o) && o.t == Value::TABLE) {
                auto ct = const_table(f, e, o.tab_ptr(), e.a->kind == Expr::Name ? e.a->str : "table");
                std::string k = emit_expr(f, *e.b), t = tmp();
                line(f, "bkv " + t + " = bk_aget(S, " + ct.first + ", " + std::to_string(ct.second) + ", " + k + ");");
                return t;
            }
            unsupported(f.chunk, e.line, "indexing this expression");
        }
        case Expr::Call: {
            std::string arr, cnt;
            emit_call(f, e, &arr, &cnt);
            if (!ad_active && last_single_of == arr && is_name(last_single)) return last_single;     // (of THIS call, not of one among its arguments)
            std::string t = tmp();
            line(f, "bkv " + t + " = " + cnt + " > 0 ? " + arr + "[0] : bk_nil();");
            if (ad_active && !ad_call.empty()) ad_note(f, t, ad_call);
            ad_call.clear();
            return t;
        }
        case Expr::Unop: {
            if (e.str == "()") return emit_expr(f, *e.a);
            if (e.str == "#") {
                int aslot = 0;
                Fn *ao = local_of(f, *e.a, &aslot);
                if (ao && ao->array_slots.count(aslot)) return num_literal((double)ao->array_slots[aslot]);
                if (ao && ao->matrix_slots.count(aslot)) return num_literal((double)ao->matrix_slots[aslot].first);
                if (ao && ao->record_slots.count(aslot)) return num_literal(0.0);                      // (no array part)
                if (e.a->kind == Expr::Index) {                                                         // #m[i]
                    int mslot = 0;
                    Fn *mo = local_of(f, *e.a->a, &mslot);
                    if (mo && mo->matrix_slots.count(mslot)) { (void)matrix_row(f, *mo, mslot, *e.a->b); return num_literal((double)mo->matrix_slots[mslot].second); }
                }
                Value sv;                                    // a constant table of the chunk (device code cannot store into one), a string constant
                if (static_value(f, *e.a, &sv) && sv.t == Value::TABLE && sv.tab()->nhash.empty() && sv.tab()->shash.empty() && !sv.tab()->meta)
                    return num_literal((double)sv.tab()->length());
                if (static_value(f, *e.a, &sv) && sv.t == Value::STR) return num_literal((double)sv.str().size());
                if (e.a->kind == Expr::String) return num_literal((double)e.a->str.size());
                unsupported(f.chunk, e.line, "the length operator on this expression");
            }
            std::string a = emit_expr(f, *e.a), t = tmp();
            if (e.str == "not") line(f, "bkv " + t + " = bk_not(" + a + ");");
            else {
                line(f, "bkv " + t + " = bk_unm(S, " + a + ");");
                if (ad_active && d_of(a) != "0.0") ad_note(f, t, "-" + d_of(a));
            }
            return t;
        }
        case Expr::Binop: {
            const std::string &op = e.str;
            if (op == "and" || op == "or") {
                std::string a = emit_expr(f, *e.a), t = tmp();
                line(f, "bkv " + t + " = " + a + ";");
                line(f, std::string("if (") + (op == "and" ? "" : "!") + "bk_truthy(" + t + ")) {");
                f.indent++;
                ++epoch;
                std::string b = emit_expr(f, *e.b);
                line(f, t + " = " + b + ";");
                f.indent--;
                line(f, "}");
                ++epoch;
                return t;
            }
            if (op == "..") unsupported(f.chunk, e.line, "string concatenation");
            std::string a = emit_expr(f, *e.a);
            std::string b = emit_expr(f, *e.b);
            std::string t = tmp(), call;
            if (op == "+") call = "bk_add(S, " + a + ", " + b + ")";
            else if (op == "-") call = "bk_sub(S, " + a + ", " + b + ")";
            else if (op == "*") call = "bk_mul(S, " + a + ", " + b + ")";
            else if (op == "/") call = "bk_div(S, " + a + ", " + b + ")";
            else if (op == "%") call = "bk_mod(S, " + a + ", " + b + ")";
            else if (op == "^") call = "bk_pow(S, " + a + ", " + b + ")";
            else if (op == "==") call = "bk_eq(S, " + a + ", " + b + ")";
            else if (op == "~=") call = "bk_ne(S, " + a + ", " + b + ")";
            else if (op == "<") call = "bk_lt(S, " + a + ", " + b + ")";
            else if (op == "<=") call = "bk_le(S, " + a + ", " + b + ")";
            else if (op == ">") call = "bk_lt(S, " + b + ", " + a + ")";
            else if (op == ">=") call = "bk_le(S, " + b + ", " + a + ")";
            else unsupported(f.chunk, e.line, "operator '" + op + "'");
            if (!ad_active) return pure_value(f, call);
            line(f, "bkv " + t + " = " + call + ";");
            if (ad_active && (d_of(a) != "0.0" || d_of(b) != "0.0")) {
                const std::string da = d_of(a), db = d_of(b);
                const bool za = da == "0.0", zb = db == "0.0";
                if (op == "+") ad_note(f, t, za ? db : zb ? da : da + " + " + db);
                else if (op == "-") ad_note(f, t, za ? "-" + db : zb ? da : da + " - " + db);
                else if (op == "*") ad_note(f, t, za ? a + ".n * " + db : zb ? b + ".n * " + da : a + ".n * " + db + " + " + b + ".n * " + da);
                else if (op == "/") ad_note(f, t, zb ? da + " / " + b + ".n" : za ? "-" + t + ".n * " + db + " / " + b + ".n" : "(" + da + " - " + t + ".n * " + db + ") / " + b + ".n");
            }
            return t;
        }
        }
        return "bk_nil()";
    }

    // evaluated argument list: fixed single values, then optionally an expanded multi-value tail
    struct Args {
        std::vector<std::string> fixed;
        bool multi = false;
        std::string marr, mcnt;
    };
    Args emit_args(Fn &f, const std::vector<ExprP> &list)
    {
        std::vector<const Expr *> ptrs;
        for (const ExprP &x : list) ptrs.push_back(x.get());
        return emit_args(f, ptrs);
    }
    Args emit_args(Fn &f, const std::vector<const Expr *> &list)
    {
        Args a;
        for (size_t i = 0; i < list.size(); ++i) {
            const Expr &x = *list[i];
            if (i + 1 == list.size() && x.kind == Expr::Call && !(ad_active && single_valued_call(f, x))) {
                a.multi = true;               // (inside a contracted loop a math call in the last place is taken as the ONE value it is:
                emit_call(f, x, &a.marr, &a.mcnt);   //  its derivative hangs on the temp emit_expr gives it)
                if (!ad_active && last_single_of == a.marr && is_name(last_single)) {   // exactly one value, and it has a name: an ordinary argument
                    a.multi = false;
                    a.fixed.push_back(last_single);
                    a.marr.clear(); a.mcnt.clear();
                }
            } else if (i + 1 == list.size() && x.kind == Expr::Vararg) {       // the extra arguments of this function, all of them
                // (value lists hold BK_MAXRET values: more extra arguments than that is this translation's limit - the script error bit)
                const std::string np = std::to_string(f.proto->nparams), vc = tmp("va");
                line(f, "int " + vc + " = na > " + np + " ? na - " + np + " : 0;");
                line(f, "if (" + vc + " > BK_MAXRET) { S.err |= BK_ERR_INDEX; " + vc + " = BK_MAXRET; }");
                a.multi = true;
                a.marr = "(a + " + np + ")";
                a.mcnt = vc;
            } else a.fixed.push_back(emit_expr(f, x));
        }
        return a;
    }
    static std::string arg_at(const Args &a, size_t i)
    {
        if (i < a.fixed.size()) return a.fixed[i];
        if (a.multi) {
            size_t k = i - a.fixed.size();
            return "(" + a.mcnt + " > " + std::to_string(k) + " ? " + a.marr + "[" + std::to_string(k) + "] : bk_nil())";
        }
        return "bk_nil()";
    }
    // materialise an evaluated value list into a fresh array; returns (array, count expression)
    std::pair<std::string, std::string> pack(Fn &f, const Args &a, int min_size)
    {
        std::string arr = tmp("v"), cnt = tmp("n");
        int cap = (int)a.fixed.size() + (a.multi ? 8 : 0);
        if (cap < min_size) cap = min_size;
        if (cap < 1) cap = 1;
        line(f, "bkv " + arr + "[" + std::to_string(cap) + "];");
        for (size_t i = 0; i < a.fixed.size(); ++i) line(f, arr + "[" + std::to_string(i) + "] = " + a.fixed[i] + ";");
        if (a.multi) {
            line(f, "int " + cnt + " = " + std::to_string(a.fixed.size()) + " + " + a.mcnt + ";");
            line(f, "for (int q = 0; q < " + a.mcnt + "; ++q) " + arr + "[" + std::to_string(a.fixed.size()) + " + q] = " + a.marr + "[q];");
        } else {
            line(f, "const int " + cnt + " = " + std::to_string(a.fixed.size()) + ";");
        }
        return {arr, cnt};
    }

    // true for calls that always produce exactly one value: math.* except modf / frexp
    bool single_valued_call(Fn &f, const Expr &e)
    {
        Value callee;
        if (e.kind != Expr::Call || !e.str.empty() || !static_value(f, *e.a, &callee) || callee.t != Value::BUILTIN) return false;
        const std::string &bn = callee.bi()->name;
        return bn.compare(0, 5, "math.") == 0 && bn != "math.modf" && bn != "math.frexp";
    }

    // a call in multi-value context: results land in *arr (bkv[BK_MAXRET]) with count *cnt
    void emit_call(Fn &f, const Expr &e, std::string *arr, std::string *cnt)
    {
        ad_call.clear();
        Value callee, self_obj;
        const bool method = !e.str.empty();
        if (method) {
            // obj:m(..) with obj a table of the chunk known now: m is looked up now, obj is bound to m's first parameter like any
            // other constant table handed to a function (below)
            if (!static_value(f, *e.a, &self_obj) || self_obj.t != Value::TABLE)
                unsupported(f.chunk, e.line, "method calls (a:" + e.str + "(..)) on a value that is not a constant table of the script");
            callee = static_field(self_obj, Value::string(e.str));
            if (callee.t != Value::FUNC) unsupported(f.chunk, e.line, "method '" + e.str + "' is not a script function");
        } else {
            int slot = 0;
            Fn *o = local_of(f, *e.a, &slot);
            if (o && o->fn_slots.count(slot)) {                      // a function defined inside this callback: a lambda of the enclosing C++ function
                if (o->fn_open.count(slot)) unsupported(f.chunk, e.line, "recursion ('" + e.a->str + "')");
                *arr = tmp("r");
                *cnt = tmp("n");
                Args a = emit_args(f, e.args);
                auto packed = pack(f, a, 1);
                line(f, "bkv " + *arr + "[BK_MAXRET];");
                line(f, "const int " + *cnt + " = " + o->fn_slots[slot] + "(" + packed.first + ", " + packed.second + ", " + *arr + ");");
                ++epoch;
                return;
            }
        }
        if (!method && !static_value(f, *e.a, &callee)) {
            std::string n = e.a->kind == Expr::Name ? "'" + e.a->str + "'" : "this expression";
            unsupported(f.chunk, e.line, "calling " + n + " (callee must be a script function or builtin known at build time)");
        }
        *arr = tmp("r");
        *cnt = tmp("n");
        if (callee.t == Value::FUNC) {
            // arguments that are script functions / builtins known now are not values on the device: the callee is generated once
            // more for this set of them, with the parameter standing for the function (its runtime argument is nil)
            std::vector<std::pair<int, Value>> bound;
            std::vector<const Expr *> plain;
            static const Expr nil_expr = [] { Expr x; x.kind = Expr::Nil; return x; }();
            const int shift = method ? 1 : 0;
            if (method) {
                if (callee.fn()->proto->nparams < 1) unsupported(f.chunk, e.line, "method '" + e.str + "' takes no self");
                bound.emplace_back(0, self_obj);
                plain.push_back(&nil_expr);
            }
            for (size_t i = 0; i < e.args.size(); ++i) {
                Value av;
                const Expr &x = *e.args[i];
                if ((int)i + shift < callee.fn()->proto->nparams && (x.kind == Expr::Name || x.kind == Expr::Index) && static_value(f, x, &av) &&
                    (av.is_function() || av.t == Value::TABLE)) {               // (a constant table of the script is bound the same way)
                    bound.emplace_back((int)i + shift, av);
                    plain.push_back(&nil_expr);
                } else plain.push_back(&x);
            }
            const FnInfo &fi = ensure_function(callee.fn(), e.line, f.chunk, bound);
            Args a = emit_args(f, plain);
            auto packed = pack(f, a, 1);
            line(f, "bkv " + *arr + "[BK_MAXRET];");
            line(f, "const int " + *cnt + " = " + fi.cname + "(S, " + packed.first + ", " + packed.second + ", " + *arr + ");");
            ++epoch;
            return;
        }
        if (callee.t != Value::BUILTIN) {
            std::string n = e.a->kind == Expr::Name ? e.a->str : "?";
            unsupported(f.chunk, e.line, "attempt to call '" + n + "' (a " + callee.type_name() + " value)");
        }
        const std::string &bn = callee.bi()->name;
        if (bn == "print") {                              // no console on the device: drop the call
            line(f, "bkv " + *arr + "[1]; const int " + *cnt + " = 0; (void)" + *arr + ";");
            return;
        }
        Args a = emit_args(f, e.args);
        auto A = [&](size_t i) { return arg_at(a, i); };
        auto single = [&](const std::string &expr) {
            line(f, "bkv " + *arr + "[1] = {" + expr + "}; const int " + *cnt + " = 1;");
            last_single = expr;
            last_single_of = *arr;
        };
        // the math library: value + error bound (bk_device_rt.h); one bkm.h call per Lua call, as in the interpreter
        static const std::map<std::string, std::string> unary = {
            {"math.sin", "bk_f_sin"}, {"math.cos", "bk_f_cos"}, {"math.tan", "bk_f_tan"}, {"math.asin", "bk_f_asin"},
            {"math.acos", "bk_f_acos"}, {"math.atan", "bk_f_atan"}, {"math.sinh", "bk_f_sinh"}, {"math.cosh", "bk_f_cosh"},
            {"math.tanh", "bk_f_tanh"}, {"math.exp", "bk_f_exp"}, {"math.log10", "bk_f_log10"}, {"math.sqrt", "bk_f_sqrt"},
            {"math.abs", "bk_f_abs"}, {"math.floor", "bk_f_floor"}, {"math.ceil", "bk_f_ceil"}};
        if (bn == "math.sin" || bn == "math.cos") {
            const std::string operand = A(0);
            auto hit = sincos.find(operand);
            if (hit == sincos.end() || hit->second.epoch != epoch) {
                SinCos sc{tmp("sn"), tmp("cs"), epoch};
                line(f, "bkv " + sc.s + ", " + sc.c + "; bk_f_sincos(S, " + operand + ", &" + sc.s + ", &" + sc.c + ");");
                hit = sincos.insert_or_assign(operand, sc).first;
            }
            single(bn == "math.sin" ? hit->second.s : hit->second.c);
            if (ad_active && d_of(operand) != "0.0")
                ad_call = bn == "math.sin" ? hit->second.c + ".n * " + d_of(operand) : "-" + hit->second.s + ".n * " + d_of(operand);
            return;
        }
        auto u = unary.find(bn);
        if (u != unary.end()) {
            if (!ad_active) { single(pure_value(f, u->second + "(S, " + A(0) + ")")); return; }
            single(u->second + "(S, " + A(0) + ")");
            if (ad_active && d_of(A(0)) != "0.0") {              // (contraction_pattern admits exactly these)
                const std::string x = A(0) + ".n", r = *arr + "[0].n", d = d_of(A(0));
                if (bn == "math.tan") ad_call = "(1.0 + " + r + " * " + r + ") * " + d;
                else if (bn == "math.asin") ad_call = d + " / bkm_sqrt((1.0 - " + x + ") * (1.0 + " + x + "))";
                else if (bn == "math.acos") ad_call = "-" + d + " / bkm_sqrt((1.0 - " + x + ") * (1.0 + " + x + "))";
                else if (bn == "math.atan") ad_call = d + " / (1.0 + " + x + " * " + x + ")";
                else if (bn == "math.sqrt") ad_call = d + " / (2.0 * " + r + ")";
                else if (bn == "math.exp") ad_call = r + " * " + d;
                else if (bn == "math.tanh") ad_call = "(1.0 - " + r + " * " + r + ") * " + d;
                else if (bn == "math.sinh") ad_call = "bkm_cosh(" + x + ") * " + d;
                else if (bn == "math.cosh") ad_call = "bkm_sinh(" + x + ") * " + d;
            }
            return;
        }
        if (bn == "math.atan2") {
            if (!ad_active) { single(pure_value(f, "bk_f_atan2(S, " + A(0) + ", " + A(1) + ")")); return; }
            single("bk_f_atan2(S, " + A(0) + ", " + A(1) + ")");
            if (ad_active && (d_of(A(0)) != "0.0" || d_of(A(1)) != "0.0")) {
                const std::string y = A(0) + ".n", x = A(1) + ".n";
                ad_call = "(" + x + " * " + d_of(A(0)) + " - " + y + " * " + d_of(A(1)) + ") / (" + x + " * " + x + " + " + y + " * " + y + ")";
            }
            return;
        }
        if (bn == "math.pow") { single("bk_powv(S, " + A(0) + ", " + A(1) + ")"); return; }
        if (bn == "math.fmod") { single("bk_f_fmod(S, " + A(0) + ", " + A(1) + ")"); return; }
        if (bn == "math.deg") { single("bk_f_scale(S, " + A(0) + ", 0x1.921fb54442d18p+1 / 180.0, true)"); return; }
        if (bn == "math.rad") { single("bk_f_scale(S, " + A(0) + ", 0x1.921fb54442d18p+1 / 180.0, false)"); return; }
        if (bn == "math.log") {
            if (a.fixed.size() < 2 && !a.multi) { single("bk_f_log(S, " + A(0) + ")"); return; }
            single("bk_f_logb(S, " + A(0) + ", " + A(1) + ")");
            return;
        }
        if (bn == "math.max" || bn == "math.min") {
            if (a.fixed.empty() && !a.multi) unsupported(f.chunk, e.line, bn + " without arguments");
            const std::string want = bn == "math.max" ? "true" : "false";
            std::string m = tmp("m");
            line(f, "bkv " + m + " = " + A(0) + "; (void)bk_tonum(S, " + m + ");");   // (no value at all -> bk_tonum(nil) raises the script error)
            for (size_t i = 1; i < a.fixed.size(); ++i) line(f, m + " = bk_f_pick(S, " + m + ", " + A(i) + ", " + want + ");");
            if (a.multi)                                             // the values a trailing call expands to, e.g. math.min(x, math.max(a, b))
                line(f, "for (int q = " + std::string(a.fixed.empty() ? "1" : "0") + "; q < " + a.mcnt + "; ++q) " + m + " = bk_f_pick(S, " + m + ", " +
                            a.marr + "[q], " + want + ");");
            single(m);
            return;
        }
        if (bn == "math.modf") {
            line(f, "bkv " + *arr + "[2]; bk_f_modf(S, " + A(0) + ", " + *arr + "); const int " + *cnt + " = 2;");
            return;
        }
        if (bn == "select") {
            // select('#', ...) counts, select(n, ...) is everything from the n-th on (n from the end when negative); the values are
            // materialised once, the index is checked as luaB_select checks it
            if (e.args.empty()) unsupported(f.chunk, e.line, "select without arguments");
            std::vector<const Expr *> rest;
            for (size_t i = 1; i < e.args.size(); ++i) rest.push_back(e.args[i].get());
            Args ra = emit_args(f, rest);
            auto packed = pack(f, ra, 1);
            if (e.args[0]->kind == Expr::String && e.args[0]->str == "#") {
                line(f, "bkv " + *arr + "[1] = {bk_num((double)" + packed.second + ")}; const int " + *cnt + " = 1; (void)" + packed.first + ";");
                return;
            }
            const std::string n = emit_expr(f, *e.args[0]), tn = tmp(), k = tmp("k");
            line(f, "const bkv " + tn + " = " + n + "; bk_need_exact(S, " + tn + ");");
            line(f, "int " + k + " = (" + tn + ".t == BK_TNUM && " + tn + ".n > -65.0 && " + tn + ".n < 65.0) ? (int)" + tn + ".n : 0;");   // (luaL_checkint truncates)
            line(f, "if (" + k + " < 0) " + k + " = " + packed.second + " + " + k + " + 1;");          // from the end
            line(f, "if (" + k + " < 1) { S.err |= BK_ERR_INDEX; " + k + " = " + packed.second + " + 1; }");     // "index out of range"
            line(f, "if (" + k + " > " + packed.second + ") " + k + " = " + packed.second + " + 1;");
            line(f, "bkv *" + *arr + " = " + packed.first + " + (" + k + " - 1); const int " + *cnt + " = " + packed.second + " - (" + k + " - 1);");
            return;
        }
        if (bn == "type") { single("bk_typeof(" + A(0) + ")"); return; }
        if (bn == "latlon_to_ray") {
            line(f, "bkv " + *arr + "[3]; const int " + *cnt + " = bk_host_latlon_to_ray(S, " + A(0) + ", " + A(1) + ", " + *arr + ");");
            return;
        }
        if (bn == "ray_to_latlon") {
            line(f, "bkv " + *arr + "[2]; const int " + *cnt + " = bk_host_ray_to_latlon(S, " + A(0) + ", " + A(1) + ", " + A(2) + ", " + *arr + ");");
            return;
        }
        if (bn == "plate_to_ray") {
            line(f, "bkv " + *arr + "[3]; const int " + *cnt + " = bk_host_plate_to_ray(S, " + A(0) + ", " + A(1) + ", " + A(2) + ", " + *arr + ");");
            return;
        }
        unsupported(f.chunk, e.line, "builtin '" + bn + "'");
    }

    // ---- the loops bk_contract applies to -----------------------------------------------------------------
    // an expression the forward-mode rules of emit_expr / emit_call cover: numbers, plain locals of this function, constants of the
    // script, + - * /, unary minus, and the smooth one- and two-argument math functions (arguments not calls themselves)
    bool ad_expr_ok(Fn &f, const Expr &e, std::vector<int> *reads)
    {
        switch (e.kind) {
        case Expr::Number: return true;
        case Expr::Name: {
            int slot = 0;
            if (Fn *o = local_of(f, e, &slot)) {
                if (o != &f || o->is_table(slot) || o->fn_slots.count(slot) || o->static_slots.count(slot) || f.proto->is_captured(slot)) return false;
                reads->push_back(slot);
                return true;
            }
            if (e.var == VarKind::Global && mutable_globals.count(e.str)) return false;
            if (e.var == VarKind::Upvalue) {
                const Scope *owner = nullptr;
                if (cell_field(upvalue_of(&f, e.slot, &owner, &slot))) return false;
            }
            Value v;
            return static_value(f, e, &v) && v.t == Value::NUM;
        }
        case Expr::Unop: return (e.str == "-" || e.str == "()") && ad_expr_ok(f, *e.a, reads);
        case Expr::Binop: return (e.str == "+" || e.str == "-" || e.str == "*" || e.str == "/") && ad_expr_ok(f, *e.a, reads) && ad_expr_ok(f, *e.b, reads);
        case Expr::Call: {
            Value callee;
            if (!e.str.empty() || !static_value(f, *e.a, &callee) || callee.t != Value::BUILTIN) return false;
            static const std::set<std::string> one = {"math.sin", "math.cos", "math.tan", "math.asin", "math.acos", "math.atan", "math.sqrt",
                                                      "math.exp", "math.sinh", "math.cosh", "math.tanh"};
            const std::string &bn = callee.bi()->name;
            const size_t want = one.count(bn) ? 1 : bn == "math.atan2" ? 2 : 0;
            if (!want || e.args.size() != want) return false;
            for (const ExprP &x : e.args)
                if (x->kind == Expr::Vararg || !ad_expr_ok(f, *x, reads)) return false;      // (a call among them is one of these: single-valued)
            return true;
        }
        default: return false;
        }
    }
    // `for i = a, b do <assignments> end`: every statement assigns plain locals of this function from ad_expr_ok expressions, and exactly
    // ONE of the assigned locals is read in a step before that step assigns it (it is carried from step to step; the others start
    // every step dead)
    bool contraction_pattern(Fn &f, const Stmt &loop, int *carried, std::vector<int> *assigned)
    {
        std::set<int> all;
        for (const StmtP &sp : loop.body) {
            const Stmt &s = *sp;
            if (s.kind == Stmt::Local) {
                if (s.exprs.size() != s.slots.size()) return false;
                for (int slot : s.slots) all.insert(slot);
            } else if (s.kind == Stmt::Assign) {
                if (s.exprs.size() != s.targets.size()) return false;
                for (const ExprP &t : s.targets) {
                    int slot = 0;
                    if (t->kind != Expr::Name || local_of(f, *t, &slot) != &f || f.is_table(slot) || f.fn_slots.count(slot) || f.static_slots.count(slot) ||
                        f.proto->is_captured(slot)) return false;
                    all.insert(slot);
                }
            } else return false;
        }
        if (all.count(loop.slots[0])) return false;                     // (the loop variable itself)
        std::set<int> done, live;
        for (const StmtP &sp : loop.body) {
            const Stmt &s = *sp;
            std::vector<int> reads;
            for (const ExprP &x : s.exprs)
                if (!ad_expr_ok(f, *x, &reads)) return false;
            for (int r : reads) if (all.count(r) && !done.count(r)) live.insert(r);
            if (s.kind == Stmt::Local) for (int slot : s.slots) done.insert(slot);
            else for (const ExprP &t : s.targets) { int slot = 0; (void)local_of(f, *t, &slot); done.insert(slot); }
        }
        if (live.size() != 1) return false;
        *carried = *live.begin();
        assigned->assign(all.begin(), all.end());
        return true;
    }

    // ---- statements ---------------------------------------------------------------------------------
    void emit_block(Fn &f, const Block &b)
    {
        ++epoch;
        for (const StmtP &s : b) emit_stmt(f, *s);
        ++epoch;
    }

    // evaluate an expression list adjusted to exactly `want` values (temps)
    std::vector<std::string> emit_values(Fn &f, const std::vector<ExprP> &exprs, size_t want)
    {
        Args a = emit_args(f, exprs);
        std::vector<std::string> v;
        for (size_t i = 0; i < want; ++i) {
            std::string t = tmp();
            line(f, "const bkv " + t + " = " + arg_at(a, i) + ";");
            if (ad_active && d_of(arg_at(a, i)) != "0.0") ad_note(f, t, d_of(arg_at(a, i)));       // (a snapshot, like the value: a, b = b, a)
            v.push_back(t);
        }
        return v;
    }

    // m[i] of a local {{..}, {..}}: a pointer expression to the row's array (indexable 1..columns).  A row that does not exist is nil in
    // Lua, and indexing nil is an error: the script error bit, as for any other one
    std::string matrix_row(Fn &f, Fn &owner, int slot, const Expr &index)
    {
        const int rows = owner.matrix_slots[slot].first, cols = owner.matrix_slots[slot].second;
        const std::string i = emit_expr(f, index), ti = tmp(), r = tmp("row");
        line(f, "const bkv " + ti + " = " + i + "; bk_need_exact(S, " + ti + ");");
        line(f, "const int " + r + " = (" + ti + ".t == BK_TNUM && " + ti + ".n >= 1.0 && " + ti + ".n <= " + std::to_string(rows) + ".0 && " + ti + ".n == bkm_trunc(" + ti + ".n)) ? (int)" + ti + ".n : 0;");
        line(f, "if (!" + r + ") S.err |= BK_ERR_INDEX;");
        return "(" + owner.ap + std::to_string(slot) + " + (" + r + " ? " + r + " - 1 : 0) * " + std::to_string(cols) + ")";
    }

    void store(Fn &f, const Expr &target, const std::string &val)
    {
        int slot = 0;
        if (target.kind == Expr::Name) {
            if (Fn *o = local_of(f, target, &slot)) {
                if (o->is_table(slot)) unsupported(f.chunk, target.line, "re-assigning table '" + target.str + "'");
                if (o->fn_slots.count(slot)) unsupported(f.chunk, target.line, "re-assigning function '" + target.str + "'");
                line(f, o->lp + std::to_string(slot) + " = " + val + ";");
                forget(o->lp + std::to_string(slot));
                if (ad_active && o == ad_fn && ad_slot.count(slot)) line(f, ad_slot[slot] + " = " + d_of(val) + ";");
            } else if (target.var == VarKind::Global) {
                line(f, "S.g_" + sanitize(target.str) + " = " + val + ";");
                forget("S.g_" + sanitize(target.str));
            } else {
                const Scope *owner = nullptr;
                const std::string *field = cell_field(upvalue_of(&f, target.slot, &owner, &slot));
                if (!field) unsupported(f.chunk, target.line, "assigning to the enclosing function's local '" + target.str + "'");
                line(f, "S." + *field + " = " + val + ";");
                forget("S." + *field);
            }
            return;
        }
        Fn *ao = local_of(f, *target.a, &slot);
        if (ao && ao->record_slots.count(slot)) {
            if (target.b->kind != Expr::String) unsupported(f.chunk, target.line, "indexing record '" + target.a->str + "' with a computed key");
            for (const std::string &n : ao->record_slots[slot]) if (n == target.b->str) { line(f, ao->field_var(slot, n) + " = " + val + ";"); forget(ao->field_var(slot, n)); return; }
            unsupported(f.chunk, target.line, "adding field '" + target.b->str + "' to '" + target.a->str + "' (name it in the constructor)");
        }
        if (target.a->kind == Expr::Index) {
            int mslot = 0;
            Fn *mo = local_of(f, *target.a->a, &mslot);
            if (mo && mo->matrix_slots.count(mslot)) {
                const std::string row = matrix_row(f, *mo, mslot, *target.a->b);
                const std::string k = emit_expr(f, *target.b);
                line(f, "bk_aset(S, " + row + ", " + std::to_string(mo->matrix_slots[mslot].second) + ", " + k + ", " + val + ");");
                return;
            }
        }
        if (ao && ao->array_slots.count(slot)) {
            std::string k = emit_expr(f, *target.b);
            line(f, "bk_aset(S, " + ao->ap + std::to_string(slot) + ", " + std::to_string(ao->array_slots[slot]) + ", " + k + ", " + val + ");");
            return;
        }
        unsupported(f.chunk, target.line, "storing into this table (only tables created by 'local t = {..}' in the same function are writable)");
    }

    // the local variables of a function: parameters from the argument array, tables as arrays, everything else nil
    static void declare_locals(std::ostream &o, const Fn &f, const std::string &pad)
    {
        const FuncProto *p = f.proto;
        for (int i = 0; i < p->nslots; ++i) {
            if (f.fn_slots.count(i)) continue;                          // (a function defined inside: declared where it is defined)
            if (f.record_slots.count(i)) {
                for (const std::string &name : f.record_slots.at(i)) o << pad << "bkv " << f.field_var(i, name) << " = bk_nil();   /* " << p->slot_names[i] << "." << name << " */\n";
            } else if (f.matrix_slots.count(i)) {
                const int cells = f.matrix_slots.at(i).first * f.matrix_slots.at(i).second;
                o << pad << "bkv " << f.ap << i << "[" << cells + 1 << "];   /* " << p->slot_names[i] << " */\n";
                o << pad << "for (int q = 0; q <= " << cells << "; ++q) " << f.ap << i << "[q] = bk_nil();\n";
            } else if (f.array_slots.count(i)) {
                // (no nil-filling: the array's size IS its constructor's length and `local t = {..}` assigns every element before anything can
                //  read one - element 0 is never addressed.  The compiler had found most of those stores dead already: quincuncial's kernel
                //  kept 12 of 40, and is no faster without them)
                o << pad << "bkv " << f.ap << i << "[" << f.array_slots.at(i) + 1 << "];   /* " << p->slot_names[i] << " */\n";
            } else if (i < p->nparams) {
                o << pad << "bkv " << f.lp << i << " = na > " << i << " ? a[" << i << "] : bk_nil();   /* " << p->slot_names[i] << " */\n";
            } else {
                o << pad << "bkv " << f.lp << i << " = bk_nil();   /* " << p->slot_names[i] << " */\n";
            }
        }
    }

    // `local function g(..) .. end` / `local g = function(..) .. end` inside device code: a lambda of the C++ function the
    // enclosing script function became, capturing by reference - which is what a Lua closure does with the locals it refers to.
    // It can be called (from the enclosing function and from functions defined after it); it is not a value.
    void emit_local_function(Fn &f, const Stmt &s)
    {
        const FuncProto *p = s.exprs[0]->proto;
        const int slot = s.slots[0];
        if (f.fn_slots.count(slot) || f.is_table(slot)) unsupported(f.chunk, s.line, "re-declaring '" + s.names[0] + "'");
        const std::string id = std::to_string(++uid);
        Fn g;
        g.proto = p;
        g.parent = &f;
        g.chunk = f.chunk;
        g.lp = "n" + id + "_l";
        g.ap = "n" + id + "_A";
        g.rp = "n" + id + "_R";
        g.indent = f.indent + 1;
        const std::string name = "NF" + id + "_" + sanitize(s.names[0]);
        f.fn_slots[slot] = name;
        f.fn_open.insert(slot);
        emit_block(g, p->body);
        f.fn_open.erase(slot);
        line(f, "/* " + f.chunk + ":" + std::to_string(p->line) + "  function " + s.names[0] + " */");
        line(f, "auto " + name + " = [&](const bkv *a, int na, bkv *r) -> int {");
        std::ostringstream decl;
        declare_locals(decl, g, std::string((size_t)(f.indent + 1) * 4, ' '));
        f.out << decl.str();
        line(f, "    (void)a; (void)na; (void)r;");
        f.out << g.out.str();
        line(f, "    return 0;");
        line(f, "};");
    }

    void emit_stmt(Fn &f, const Stmt &s)
    {
        // (plain assignments leave the sin / cos memory alone - store() forgets what they assign; everything else opens scopes or repeats)
        const bool plain = s.kind == Stmt::Local || s.kind == Stmt::Assign;
        if (!plain) ++epoch;
        emit_stmt_body(f, s);
        if (!plain) ++epoch;
    }
    void emit_stmt_body(Fn &f, const Stmt &s)
    {
        switch (s.kind) {
        case Stmt::Local: {
            if (s.slots.size() == 1 && s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Function) { emit_local_function(f, s); return; }
            if (s.slots.size() == 1 && s.exprs.size() == 1 && (s.exprs[0]->kind == Expr::Name || s.exprs[0]->kind == Expr::Index)) {
                Value fv;                                              // `local s = math.sin`, `local g = helper`: a name for that function
                if (static_value(f, *s.exprs[0], &fv) && (fv.is_function() || fv.t == Value::TABLE)) {
                    if (!slot_is_constant(f.proto, s.slots[0])) unsupported(f.chunk, s.line, std::string(fv.t == Value::TABLE ? "table" : "function") + " '" + s.names[0] + "' used as a value (the local is assigned or captured later)");
                    f.static_slots[s.slots[0]] = fv;
                    return;
                }
            }
            if (s.slots.size() == 1 && s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Table && !s.exprs[0]->fields.empty() && s.exprs[0]->args.empty()) {
                // local p = {x = .., y = ..}: a record - every field a variable of the C++ function
                const Expr &t = *s.exprs[0];
                const int slot = s.slots[0];
                if (f.fn_slots.count(slot) || f.is_table(slot)) unsupported(f.chunk, s.line, "re-declaring a table");
                std::vector<std::string> names, vals;
                for (auto &fld : t.fields) {
                    if (fld.first->kind != Expr::String || fld.first->str.empty() || !(isalpha((unsigned char)fld.first->str[0]) || fld.first->str[0] == '_'))
                        unsupported(f.chunk, s.line, "table constructors with computed keys");
                    for (const std::string &n : names) if (n == fld.first->str) unsupported(f.chunk, s.line, "field '" + n + "' named twice in a table constructor");
                    names.push_back(fld.first->str);
                    vals.push_back(emit_expr(f, *fld.second));
                }
                f.record_slots[slot] = names;
                for (size_t i = 0; i < names.size(); ++i) { line(f, f.field_var(slot, names[i]) + " = " + vals[i] + ";"); forget(f.field_var(slot, names[i])); }
                return;
            }
            if (s.slots.size() == 1 && s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Table && s.exprs[0]->fields.empty() && !s.exprs[0]->args.empty() &&
                s.exprs[0]->args[0]->kind == Expr::Table) {
                // local m = {{a, b}, {c, d}}: rows of one length, kept in one flat array
                const Expr &t = *s.exprs[0];
                const int slot = s.slots[0];
                if (f.fn_slots.count(slot) || f.is_table(slot)) unsupported(f.chunk, s.line, "re-declaring a table");
                const int rows = (int)t.args.size(), cols = (int)t.args[0]->args.size();
                std::vector<std::string> vals;
                for (auto &row : t.args) {
                    if (row->kind != Expr::Table || !row->fields.empty() || (int)row->args.size() != cols || cols == 0)
                        unsupported(f.chunk, s.line, "nested table constructors other than rows of one length ({{a, b}, {c, d}})");
                    if (row->args.back()->kind == Expr::Call && !single_valued_call(f, *row->args.back()))
                        unsupported(f.chunk, s.line, "call expansion inside a table constructor (write '(f(...))' to keep one value)");
                    if (row->args.back()->kind == Expr::Vararg)
                        unsupported(f.chunk, s.line, "'...' at the end of a table constructor (a table of a size only known at run time)");
                    for (auto &x : row->args) vals.push_back(emit_expr(f, *x));
                }
                f.matrix_slots[slot] = {rows, cols};
                for (size_t i = 0; i < vals.size(); ++i) line(f, f.ap + std::to_string(slot) + "[" + std::to_string(i + 1) + "] = " + vals[i] + ";");
                return;
            }
            if (s.slots.size() == 1 && s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Table) {
                const Expr &t = *s.exprs[0];
                if (!t.fields.empty()) unsupported(f.chunk, s.line, "table constructors mixing positional and named fields");
                // a trailing call would expand to all its results; the table's size must be static, so only calls that
                // always yield exactly one value are taken (the math library; `(f())` truncates any other call)
                if (!t.args.empty() && t.args.back()->kind == Expr::Call && !single_valued_call(f, *t.args.back()))
                    unsupported(f.chunk, s.line, "call expansion inside a table constructor (write '(f(...))' to keep one value)");
                if (!t.args.empty() && t.args.back()->kind == Expr::Vararg)
                    unsupported(f.chunk, s.line, "'...' at the end of a table constructor (a table of a size only known at run time)");
                std::vector<std::string> vals;
                for (auto &x : t.args) vals.push_back(emit_expr(f, *x));
                int n = (int)vals.size();
                if (f.array_slots.count(s.slots[0]) && f.array_slots[s.slots[0]] != n) unsupported(f.chunk, s.line, "re-declaring a table with a different size");
                if (f.record_slots.count(s.slots[0]) || f.matrix_slots.count(s.slots[0])) unsupported(f.chunk, s.line, "re-declaring a table");
                f.array_slots[s.slots[0]] = n;
                for (int i = 0; i < n; ++i) line(f, f.ap + std::to_string(s.slots[0]) + "[" + std::to_string(i + 1) + "] = " + vals[i] + ";");
                return;
            }
            auto v = emit_values(f, s.exprs, s.slots.size());
            for (size_t i = 0; i < s.slots.size(); ++i) {
                line(f, f.lp + std::to_string(s.slots[i]) + " = " + v[i] + ";");
                forget(f.lp + std::to_string(s.slots[i]));
                if (ad_active && &f == ad_fn && ad_slot.count(s.slots[i])) line(f, ad_slot[s.slots[i]] + " = " + d_of(v[i]) + ";");
            }
            return;
        }
        case Stmt::LocalFunction: emit_local_function(f, s); return;
        case Stmt::Assign: {
            auto v = emit_values(f, s.exprs, s.targets.size());
            for (size_t i = 0; i < s.targets.size(); ++i) store(f, *s.targets[i], v[i]);
            return;
        }
        case Stmt::CallStmt: {
            std::string arr, cnt;
            line(f, "{");
            f.indent++;
            emit_call(f, *s.call, &arr, &cnt);
            line(f, "(void)" + cnt + ";");
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::Do:
            line(f, "{");
            f.indent++;
            emit_block(f, s.body);
            f.indent--;
            line(f, "}");
            return;
        case Stmt::While: {
            line(f, "for (;;) {");
            f.indent++;
            line(f, "if (!bk_tick(S)) break;");
            std::string c = emit_expr(f, *s.cond);
            line(f, "if (!bk_truthy(" + c + ")) break;");
            emit_block(f, s.body);
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::Repeat: {
            line(f, "for (;;) {");
            f.indent++;
            line(f, "if (!bk_tick(S)) break;");
            emit_block(f, s.body);
            std::string c = emit_expr(f, *s.cond);
            line(f, "if (bk_truthy(" + c + ")) break;");
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::If: {
            int opened = 0;
            for (size_t i = 0; i < s.clauses.size(); ++i) {
                const auto &c = s.clauses[i];
                if (!c.first) {
                    emit_block(f, c.second);
                    break;
                }
                std::string cv = emit_expr(f, *c.first);
                line(f, "if (bk_truthy(" + cv + ")) {");
                f.indent++;
                emit_block(f, c.second);
                f.indent--;
                if (i + 1 < s.clauses.size()) {
                    line(f, "} else {");
                    f.indent++;
                    opened++;
                } else line(f, "}");
            }
            for (int k = 0; k < opened; ++k) { f.indent--; line(f, "}"); }
            return;
        }
        case Stmt::NumFor: {
            std::string start = emit_expr(f, *s.exprs[0]), stop = emit_expr(f, *s.exprs[1]);
            std::string step = s.exprs.size() > 2 ? emit_expr(f, *s.exprs[2]) : "bk_num(1.0)";
            std::string st = tmp("step"), lim = tmp("lim"), idx = tmp("idx");
            line(f, "bk_need_exact(S, " + start + "); bk_need_exact(S, " + stop + "); bk_need_exact(S, " + step + ");");   // loop trip count
            line(f, "const double " + st + " = bk_tonum(S, " + step + "), " + lim + " = bk_tonum(S, " + stop + ");");
            line(f, "double " + idx + " = bk_tonum(S, " + start + ") - " + st + ";");       // OP_FORPREP
            line(f, "for (;;) {");
            f.indent++;
            line(f, idx + " = " + idx + " + " + st + ";");                                     // OP_FORLOOP
            line(f, "if (!(0 < " + st + " ? " + idx + " <= " + lim + " : " + lim + " <= " + idx + ")) break;");
            line(f, "if (!bk_tick(S)) break;");
            line(f, f.lp + std::to_string(s.slots[0]) + " = bk_num(" + idx + ");");
            int carried = -1;
            std::vector<int> assigned;
            const bool contract = !ad_active && contraction_pattern(f, s, &carried, &assigned);
            std::string e0;
            if (contract) {
                const std::string k = std::to_string(++uid), cv = f.lp + std::to_string(carried);
                e0 = "ce" + k;
                line(f, "const double " + e0 + " = " + cv + ".e; " + cv + ".e = 0.0;     /* " + f.proto->slot_names[(size_t)carried] + " enters the step as exact (bk_contract below) */");
                for (int slot : assigned) {
                    const std::string dn = "cd" + k + "_" + std::to_string(slot);
                    line(f, "double " + dn + " = " + (slot == carried ? "1.0" : "0.0") + ";");
                    ad_slot[slot] = dn;
                    ad_d[f.lp + std::to_string(slot)] = dn;
                }
                ad_active = true;
                ad_fn = &f;
            }
            emit_block(f, s.body);
            if (contract) {
                ad_active = false;
                for (int slot : assigned)
                    line(f, f.lp + std::to_string(slot) + ".e = bk_contract(S, " + ad_slot[slot] + ", " + e0 + ", " + f.lp + std::to_string(slot) + ".e);");
                ad_slot.clear();
                ad_d.clear();
                ad_fn = nullptr;
            }
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::GenFor: {
            // `for i, v in ipairs(t)` / `for k, v in pairs(t)` over a table whose size is known when the code is generated:
            // one made by `local t = {..}` in the same function, or a global / upvalue array table nobody assigns
            const Expr *call = s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Call ? s.exprs[0].get() : nullptr;
            Value callee;
            if (!call || !static_value(f, *call->a, &callee) || callee.t != Value::BUILTIN ||
                (callee.bi()->name != "ipairs" && callee.bi()->name != "pairs") || call->args.size() != 1)
                unsupported(f.chunk, s.line, "generic 'for ... in' other than ipairs(t) / pairs(t)");
            const bool is_ipairs = callee.bi()->name == "ipairs";
            const Expr &targ = *call->args[0];
            std::string arr;
            int n = 0;
            int tslot = 0;
            Fn *to = local_of(f, targ, &tslot);
            if (to && to->array_slots.count(tslot)) {
                arr = to->ap + std::to_string(tslot);
                n = to->array_slots[tslot];
            } else {
                Value tv;
                if (!static_value(f, targ, &tv) || tv.t != Value::TABLE) unsupported(f.chunk, s.line, "iterating a table that is not known when the kernel is generated");
                auto ct = const_table(f, targ, tv.tab_ptr(), targ.kind == Expr::Name ? targ.str : "table");
                arr = ct.first;
                n = ct.second;
            }
            std::string gi = tmp("gi"), gv = tmp("gv");
            line(f, "for (int " + gi + " = 1; " + gi + " <= " + std::to_string(n) + "; ++" + gi + ") {");
            f.indent++;
            line(f, "if (!bk_tick(S)) break;");
            line(f, "const bkv " + gv + " = " + arr + "[" + gi + "];");
            line(f, std::string("if (") + gv + ".t == BK_TNIL) " + (is_ipairs ? "break;" : "continue;"));   // ipairs stops at the first nil, pairs skips it
            for (size_t i = 0; i < s.slots.size(); ++i)
                line(f, f.lp + std::to_string(s.slots[i]) + " = " + (i == 0 ? "bk_num((double)" + gi + ")" : i == 1 ? gv : std::string("bk_nil()")) + ";");
            emit_block(f, s.body);
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::Return: {
            if (s.exprs.size() == 1 && s.exprs[0]->kind == Expr::Call) {
                std::string arr, cnt;
                line(f, "{");
                f.indent++;
                emit_call(f, *s.exprs[0], &arr, &cnt);
                line(f, "for (int q = 0; q < " + cnt + "; ++q) r[q] = " + arr + "[q];");
                line(f, "return " + cnt + ";");
                f.indent--;
                line(f, "}");
                return;
            }
            line(f, "{");
            f.indent++;
            Args a = emit_args(f, s.exprs);
            if (a.fixed.size() > 8 || (a.multi && a.fixed.size() > 4)) unsupported(f.chunk, s.line, "more than 8 return values");
            for (size_t i = 0; i < a.fixed.size(); ++i) line(f, "r[" + std::to_string(i) + "] = " + a.fixed[i] + ";");
            if (a.multi) {
                line(f, "for (int q = 0; q < " + a.mcnt + " && " + std::to_string(a.fixed.size()) + " + q < BK_MAXRET; ++q) r[" + std::to_string(a.fixed.size()) + " + q] = " + a.marr + "[q];");
                line(f, "return " + std::to_string(a.fixed.size()) + " + " + a.mcnt + ";");
            } else line(f, "return " + std::to_string(a.fixed.size()) + ";");
            f.indent--;
            line(f, "}");
            return;
        }
        case Stmt::Break: line(f, "break;"); return;
        case Stmt::Goto: case Stmt::Label: unsupported(f.chunk, s.line, "goto");
        }
    }

    // ---- functions -------------------------------------------------------------------------------------
    const FnInfo &ensure_function(const Closure *cl, int line_no, const std::string &from_chunk, const std::vector<std::pair<int, Value>> &bound = {})
    {
        std::string signature;
        for (auto &b : bound) {
            char id[48];
            snprintf(id, sizeof id, "%d=%p;", b.first, b.second.p.get());
            signature += id;
        }
        FnInfo &fi = fns[{cl->proto, signature}];
        if (fi.done) {
            if (fi.cl != cl) unsupported(from_chunk, line_no, "two closures of the same function body");
            return fi;
        }
        if (fi.emitting) unsupported(from_chunk, line_no, "recursion ('" + cl->proto->name + "')");
        fi.proto = cl->proto;
        fi.cl = cl;
        fi.cname = "LF" + std::to_string(++uid) + "_" + sanitize(cl->proto->name);
        fi.emitting = true;

        Fn f;
        f.proto = cl->proto;
        f.cl = cl;
        f.chunk = cl->chunk->name;
        for (auto &b : bound) {
            if (!slot_is_constant(cl->proto, b.first))
                unsupported(from_chunk, line_no, std::string(b.second.t == Value::TABLE ? "a table" : "a function") + " passed as '" + cl->proto->slot_names[(size_t)b.first] + "' of '" + cl->proto->name + "', which assigns or captures that parameter");
            f.static_slots[b.first] = b.second;
        }
        emit_block(f, cl->proto->body);

        std::ostringstream o;
        o << "/* " << f.chunk << ":" << cl->proto->line << "  function " << cl->proto->name << " */\n";
        o << "BK_DEV int " << fi.cname << "(BkState &S, const bkv *a, int na, bkv *r)\n{\n";
        declare_locals(o, f, "    ");
        o << "    (void)a; (void)na; (void)r;\n";
        o << f.out.str();
        o << "    return 0;\n}\n\n";
        fn_code.push_back(o.str());
        fi.emitting = false;
        fi.done = true;
        return fi;
    }
};

}  // namespace


namespace {

// Does a callback read a script global that callbacks assign before assigning it itself - state carried from pixel to pixel in the
// reference's one sequential scan (fisheye.c:2084-2124), fresh per pixel on the GPU?  A definite-assignment walk over the callbacks
// and what they call, with ONE carried read recognised as harmless: the KEYED CACHE (eckert4.lua:14-21)
//     if P ~= K then  G1 = ..; G2 = ..;  K = P  end          -- then G1, G2 are read
// where P is one of the callback's own parameters handed down unchanged, everything stored in the branch is computed from that
// parameter alone (no other parameter, no carried state), K and the G's are stored nowhere else, and K does not start out as a
// number.  Then whatever the G's hold after the `if` is what the branch would compute for this pixel: taken or not, carried over or
// fresh, the pixel's result is the same - not state.  Every value therefore carries which callback parameters it depends on
// (bits 0..2; IMPURE = carried state or something the walk cannot see through) and whether it IS such a parameter, unchanged.
struct StateScan {
    Interp &I;
    const std::set<std::string> &mut;
    std::string culprit;
    std::set<const FuncProto *> active;
    static constexpr uint8_t IMPURE = 0x80;
    struct Val { uint8_t dep = 0; int copy_of = -1; };
    using State = std::map<std::string, Val>;            // the mutable globals definitely assigned on this path, and what they hold
    struct Frame { std::vector<Val> loc; uint8_t ret = 0; };
    std::map<std::string, std::string> cache_key;        // a cache global (the key global included) -> its key global
    std::string refresh_key;                             // walking the branch of `if P ~= <refresh_key> then`
    std::vector<std::pair<std::string, std::string>> stores;      // every store to a mutable global: (name, refresh_key it was made under)

    static Val join(Val a, Val b) { Val o; o.dep = (uint8_t)(a.dep | b.dep); o.copy_of = a.copy_of == b.copy_of ? a.copy_of : -1; return o; }
    static State meet(const State &a, const State &b)
    {
        State o;
        for (auto &kv : a) { auto it = b.find(kv.first); if (it != b.end()) o[kv.first] = join(kv.second, it->second); }
        return o;
    }
    Val local_of(Frame &fr, int slot) { if (slot < 0) return Val{IMPURE, -1}; if ((size_t)slot >= fr.loc.size()) fr.loc.resize((size_t)slot + 1); return fr.loc[(size_t)slot]; }
    void set_local(Frame &fr, int slot, Val v, bool fresh)
    {
        if (slot < 0) return;
        if ((size_t)slot >= fr.loc.size()) fr.loc.resize((size_t)slot + 1);
        fr.loc[(size_t)slot] = fresh ? v : join(fr.loc[(size_t)slot], v);       // (a re-assignment may sit in a branch or a loop: never less than it was)
    }
    Val read_global(const std::string &name, const State &st)
    {
        if (!mut.count(name)) return Val{};                                    // nobody assigns it: a constant of the script
        auto it = st.find(name);
        if (it != st.end()) return it->second;
        if (culprit.empty()) culprit = name;
        return Val{IMPURE, -1};
    }
    Val call_script(const Closure *cl, const std::vector<Val> &args, const State &st)
    {
        if (active.count(cl->proto)) return Val{IMPURE, -1};                    // (recursion is refused by the code generator anyway)
        active.insert(cl->proto);
        Frame fr;
        fr.loc.resize((size_t)std::max(cl->proto->nslots, cl->proto->nparams));
        for (int i = 0; i < cl->proto->nparams; ++i) fr.loc[(size_t)i] = (size_t)i < args.size() ? args[(size_t)i] : Val{};
        if (cl->proto->is_vararg) for (size_t i = (size_t)cl->proto->nparams; i < args.size(); ++i) fr.ret |= args[i].dep;      // (`...` may come back out)
        Emitter::Scope callee_scope;
        callee_scope.proto = cl->proto;
        callee_scope.cl = cl;
        State inner = st;                                    // the callee sees what is assigned so far; its own stores stay its own
        block(cl->proto->body, &callee_scope, inner, fr);
        active.erase(cl->proto);
        Val out;
        out.dep = fr.ret;
        return out;
    }
    Val expr(const Expr *e, const Emitter::Scope *sc, State &st, Frame &fr)
    {
        if (!e) return Val{};
        switch (e->kind) {
        case Expr::Nil: case Expr::True: case Expr::False: case Expr::Number: case Expr::String: return Val{};
        case Expr::Vararg: return Val{fr.ret, -1};           // (what came in through `...` was folded into ret by call_script)
        case Expr::Name:
            if (e->var == VarKind::Global) return read_global(e->str, st);
            if (e->var == VarKind::Local) return local_of(fr, e->slot);
            return Val{};                                    // an upvalue: callbacks that assign one count as carrying state before this walk starts
        case Expr::Function:
            // a function defined inside device code can only be called after this point, where at least as much is assigned as here:
            // its body is walked once, now, as if it were called here (its own stores stay its own)
            if (e->proto) {
                Emitter::Scope inner_scope;
                inner_scope.proto = e->proto;
                inner_scope.parent = sc;
                State inner = st;
                Frame ifr;
                ifr.loc.assign((size_t)e->proto->nslots, Val{IMPURE, -1});      // (its parameters: whatever it will be called with)
                block(e->proto->body, &inner_scope, inner, ifr);
            }
            return Val{IMPURE, -1};
        default: break;
        }
        if (e->kind == Expr::Call) {
            std::vector<Val> args;
            Val all;
            for (auto &x : e->args) { args.push_back(expr(x.get(), sc, st, fr)); all.dep |= args.back().dep; }
            Value callee;
            bool known = false;
            if (!e->str.empty()) {                                               // obj:m(..): m's body is walked like any callee's
                expr(e->a.get(), sc, st, fr);
                Value obj;
                if (e->a->kind == Expr::Name && e->a->var == VarKind::Global && !mut.count(e->a->str)) obj = I.get_global(e->a->str);
                else if (e->a->kind == Expr::Name && e->a->var == VarKind::Upvalue) {
                    const Emitter::Scope *owner = nullptr;
                    int slot = 0;
                    if (const Value *cell = Emitter::upvalue_of(sc, e->a->slot, &owner, &slot)) obj = *cell;
                }
                callee = Emitter::static_field(obj, Value::string(e->str));
                known = callee.t == Value::FUNC;
                if (known) args.insert(args.begin(), Val{});                     // self: a constant table
            } else if (e->a->kind == Expr::Name && e->a->var == VarKind::Global && !mut.count(e->a->str)) { callee = I.get_global(e->a->str); known = true; }
            else if (e->a->kind == Expr::Name && e->a->var == VarKind::Upvalue) {
                const Emitter::Scope *owner = nullptr;
                int slot = 0;
                if (const Value *cell = Emitter::upvalue_of(sc, e->a->slot, &owner, &slot)) { callee = *cell; known = true; }
            } else {
                const Val f = expr(e->a.get(), sc, st, fr);                      // math.sin, lib.f, a function held in a local ...
                if (e->a->kind == Expr::Index) { all.dep |= f.dep; return Val{all.dep, -1}; }      // a field of a table: builtins and constant tables' functions are pure in their arguments
                return Val{IMPURE, -1};                                          // a function VALUE (parameter, local): not followed
            }
            if (known && callee.t == Value::FUNC) return call_script(callee.fn(), args, st);
            return Val{all.dep, -1};                                             // a builtin: a pure function of its arguments
        }
        Val out;
        out = join(out, expr(e->a.get(), sc, st, fr));
        if (e->kind == Expr::Unop && e->op == Expr::OP_PAREN) return out;        // (x) is still x
        out.copy_of = -1;
        if (e->b) out.dep |= expr(e->b.get(), sc, st, fr).dep;
        for (auto &x : e->args) out.dep |= expr(x.get(), sc, st, fr).dep;
        for (auto &x : e->fields) { out.dep |= expr(x.first.get(), sc, st, fr).dep; out.dep |= expr(x.second.get(), sc, st, fr).dep; }
        return out;
    }
    // `if P ~= K then .. end` as the refresh of a keyed cache (see above); true = handled, st updated
    bool keyed_refresh(const Stmt &s, const Emitter::Scope *sc, State &st, Frame &fr)
    {
        if (s.clauses.size() != 1 || !s.clauses[0].first || !refresh_key.empty()) return false;
        const Expr *c = s.clauses[0].first.get();
        if (c->kind != Expr::Binop || c->op != Expr::OP_NE) return false;
        const Expr *k = c->b.get(), *p = c->a.get();
        auto is_key = [&](const Expr *x) { return x && x->kind == Expr::Name && x->var == VarKind::Global && mut.count(x->str) && !st.count(x->str); };
        if (!is_key(k)) std::swap(k, p);
        if (!is_key(k) || !p || p->kind != Expr::Name || p->var != VarKind::Local) return false;
        const Val pv = local_of(fr, p->slot);
        if (pv.copy_of < 0 || pv.dep != (uint8_t)(1u << pv.copy_of)) return false;             // the key must BE a callback parameter
        if (I.get_global(k->str).t == Value::NUM) return false;                                  // (a key that starts out as a number could match the first pixel)
        const std::string saved_culprit = culprit;
        State body = st;
        Frame bfr = fr;
        refresh_key = k->str;
        const size_t stores_before = stores.size();
        const bool falls = block(s.clauses[0].second, sc, body, bfr);
        refresh_key.clear();
        bool ok = falls && culprit == saved_culprit;
        auto kv = body.find(k->str);
        ok = ok && kv != body.end() && kv->second.copy_of == pv.copy_of;                        // K = P on every way through the branch
        std::vector<std::string> fresh;
        for (auto &g : body) if (!st.count(g.first)) { fresh.push_back(g.first); ok = ok && (g.second.dep & ~(uint8_t)(1u << pv.copy_of)) == 0; }
        for (size_t i = stores_before; ok && i < stores.size(); ++i) ok = body.count(stores[i].first) != 0;      // every store of the branch is a definite one
        if (!ok) {                                           // not that pattern: undo, let the ordinary rules speak
            culprit = saved_culprit;
            stores.resize(stores_before);
            return false;
        }
        for (const std::string &g : fresh) {
            auto had = cache_key.find(g);
            if (had != cache_key.end() && had->second != k->str && culprit.empty()) culprit = g;      // two keys for one cache
            cache_key[g] = k->str;
            st[g] = body[g];
        }
        fr.loc = bfr.loc;                                    // (locals the branch re-assigned: joined, never less than they were)
        return true;
    }
    // returns whether control can fall out of the end of the block
    bool block(const Block &b, const Emitter::Scope *sc, State &st, Frame &fr)
    {
        for (const StmtP &sp : b) {
            const Stmt &s = *sp;
            switch (s.kind) {
            case Stmt::Return:
                for (auto &x : s.exprs) fr.ret |= expr(x.get(), sc, st, fr).dep;
                return false;
            case Stmt::Break: return false;
            case Stmt::Goto: case Stmt::Label: break;        // (refused by the code generator)
            case Stmt::If: {
                if (keyed_refresh(s, sc, st, fr)) break;
                State met;
                bool any = false, has_else = false;
                for (auto &c : s.clauses) {
                    if (c.first) expr(c.first.get(), sc, st, fr); else has_else = true;
                    State d = st;
                    if (block(c.second, sc, d, fr)) { met = any ? meet(met, d) : d; any = true; }
                }
                if (!has_else) { met = any ? meet(met, st) : st; any = true; }
                if (!any) return false;                               // every branch returned
                st = met;
                break;
            }
            case Stmt::While: case Stmt::NumFor: case Stmt::GenFor: {
                Val ctl;
                if (s.cond) ctl.dep |= expr(s.cond.get(), sc, st, fr).dep;
                for (auto &x : s.exprs) ctl.dep |= expr(x.get(), sc, st, fr).dep;
                for (int slot : s.slots) set_local(fr, slot, Val{ctl.dep, -1}, true);
                for (int pass = 0; pass < 3; ++pass) {                // (what a local depends on can grow from one trip to the next)
                    State d = st;
                    block(s.body, sc, d, fr);                         // may run zero times: nothing it stores is definite afterwards
                    if (s.cond) expr(s.cond.get(), sc, d, fr);
                }
                break;
            }
            case Stmt::Repeat: {
                State d = st;
                bool falls = true;
                for (int pass = 0; pass < 3; ++pass) { d = st; falls = block(s.body, sc, d, fr); expr(s.cond.get(), sc, d, fr); }
                if (falls) st = d;                                    // the body runs at least once
                break;
            }
            case Stmt::Do: if (!block(s.body, sc, st, fr)) return false; break;
            case Stmt::LocalFunction:
                for (int slot : s.slots) set_local(fr, slot, Val{}, true);
                for (auto &x : s.exprs) expr(x.get(), sc, st, fr);
                break;
            case Stmt::Local: {
                std::vector<Val> vals;
                for (auto &x : s.exprs) vals.push_back(expr(x.get(), sc, st, fr));
                for (size_t i = 0; i < s.slots.size(); ++i) {
                    Val v;                                            // (no initialiser: nil)
                    if (i < vals.size()) v = vals[i];
                    else if (!vals.empty() && s.exprs.back()->kind == Expr::Call) v = Val{vals.back().dep, -1};     // a call's further results
                    if (i + 1 >= vals.size() && !vals.empty() && s.exprs.back()->kind == Expr::Call) v.copy_of = -1;
                    set_local(fr, s.slots[i], v, true);
                }
                break;
            }
            default: {
                std::vector<Val> vals;
                for (auto &x : s.exprs) vals.push_back(expr(x.get(), sc, st, fr));       // right-hand sides first ...
                expr(s.call.get(), sc, st, fr);
                for (size_t i = 0; i < s.targets.size(); ++i) {
                    const Expr *t = s.targets[i].get();
                    Val v;
                    if (i < vals.size()) v = vals[i];
                    else if (!vals.empty() && s.exprs.back()->kind == Expr::Call) v = Val{vals.back().dep, -1};
                    if (i + 1 >= vals.size() && !vals.empty() && s.exprs.back()->kind == Expr::Call) v.copy_of = -1;
                    if (t->kind == Expr::Name && t->var == VarKind::Global) {           // ... then the store
                        if (mut.count(t->str)) { st[t->str] = v; stores.emplace_back(t->str, refresh_key); }
                    } else if (t->kind == Expr::Name && t->var == VarKind::Local) set_local(fr, t->slot, v, false);
                    else if (t->kind == Expr::Index) { expr(t->a.get(), sc, st, fr); expr(t->b.get(), sc, st, fr); }
                }
                break;
            }
            }
        }
        return true;
    }
    // after every callback has been walked: a cache global stored anywhere but in its own refresh branch is state after all
    void check_caches()
    {
        for (auto &sv : stores) {
            auto c = cache_key.find(sv.first);
            if (c != cache_key.end() && sv.second != c->second && culprit.empty()) culprit = sv.first;
        }
    }
};

}  // namespace

bool callbacks_carry_state(const EmitRequest &req, std::string *which)
{
    Emitter em(*req.interp);
    std::set<const FuncProto *> seen;
    const Value *roots[3] = {&req.lens_inverse, &req.lens_forward, &req.globe_plate};
    for (const Value *v : roots)
        if (v->t == Value::FUNC && !seen.count(v->fn()->proto)) {
            seen.insert(v->fn()->proto);
            em.scan_closure(v->fn(), seen);
        }
    if (!em.mutable_cells.empty()) {
        // a local of the script that callbacks assign: not followed path by path like the globals - taken to carry state
        if (which) *which = em.mutable_cells[0].second.substr(em.mutable_cells[0].second.find('_') + 1);
        return true;
    }
    if (em.mutable_globals.empty()) return false;
    StateScan sc{*req.interp, em.mutable_globals, std::string(), {}, {}, std::string(), {}};
    for (const Value *v : roots)
        if (v->t == Value::FUNC) {
            StateScan::State st;
            StateScan::Frame fr;
            const FuncProto *pr = v->fn()->proto;
            fr.loc.resize((size_t)std::max(pr->nslots, pr->nparams));
            for (int i = 0; i < pr->nparams && i < 3; ++i) fr.loc[(size_t)i] = StateScan::Val{(uint8_t)(1u << i), i};       // the callback's own parameters
            for (int i = 3; i < pr->nparams; ++i) fr.loc[(size_t)i] = StateScan::Val{};
            Emitter::Scope root;
            root.proto = pr;
            root.cl = v->fn();
            sc.active.insert(pr);
            sc.block(pr->body, &root, st, fr);
            sc.active.erase(pr);
        }
    sc.check_caches();
    if (which) *which = sc.culprit;
    return !sc.culprit.empty();
}

std::string emit_build_source(const EmitRequest &req)
{
    Emitter em(*req.interp);
    std::set<const FuncProto *> seen;
    const Value *roots[3] = {&req.lens_inverse, &req.lens_forward, &req.globe_plate};
    for (const Value *v : roots)
        if (v->t == Value::FUNC && !seen.count(v->fn()->proto)) {
            seen.insert(v->fn()->proto);
            em.scan_closure(v->fn(), seen);
        }
    for (const Value *v : roots)
        if (v->t != Value::NIL && v->t != Value::FUNC)
            throw LuaError(std::string("a lens / globe callback must be a Lua function, got a ") + v->type_name());

    std::string names[3];
    for (int i = 0; i < 3; ++i)
        if (roots[i]->t == Value::FUNC) names[i] = em.ensure_function(roots[i]->fn(), 0, "callback").cname;

    std::ostringstream src;
    src << "/* generated by libblinkyhip (bk_emit.cpp) */\n";
    src << "#define BK_MUTABLE_GLOBALS";
    for (const std::string &g : em.mutable_globals) src << " bkv g_" << sanitize(g) << ";";
    for (auto &c : em.mutable_cells) src << " bkv " << c.second << ";";
    src << "\n";
    src << "#include \"bkm.h\"\n#include \"bk_build_params.h\"\n#include \"bk_device_rt.h\"\n\n";
    src << em.table_code.str() << "\n";
    for (const std::string &c : em.fn_code) src << c;
    src << "#define BK_INIT_GLOBALS(S)";
    for (const std::string &g : em.mutable_globals) {
        Value v = req.interp->get_global(g);
        std::string init;
        switch (v.t) {
        case Value::NIL: init = "bk_nil()"; break;
        case Value::BOOL: init = v.b ? "bk_bool(true)" : "bk_bool(false)"; break;
        case Value::NUM: init = num_literal(v.n); break;
        case Value::STR: init = em.str_literal(v.str()); break;
        default:
            throw LuaError("global '" + g + "' is assigned inside a GPU callback but holds a " + v.type_name() +
                           " when the lensmap build starts; only nil / boolean / number / string globals can be per-pixel state");
        }
        src << " (S).g_" << sanitize(g) << " = " << init << ";";
    }
    for (auto &c : em.mutable_cells) {                               // chunk locals the callbacks assign: the same per-thread treatment
        const Value &v = *c.first;
        std::string init;
        switch (v.t) {
        case Value::NIL: init = "bk_nil()"; break;
        case Value::BOOL: init = v.b ? "bk_bool(true)" : "bk_bool(false)"; break;
        case Value::NUM: init = num_literal(v.n); break;
        case Value::STR: init = em.str_literal(v.str()); break;
        default:
            throw LuaError("local '" + c.second.substr(c.second.find('_') + 1) + "' of the script is assigned inside a GPU callback but holds a " +
                           v.type_name() + " when the lensmap build starts; only nil / boolean / number / string values can be per-pixel state");
        }
        src << " (S)." << c.second << " = " << init << ";";
    }
    src << "\n";
    if (!names[0].empty()) src << "#define BK_HAS_INVERSE 1\n#define LF_lens_inverse " << names[0] << "\n";
    if (!names[1].empty()) src << "#define BK_HAS_FORWARD 1\n#define LF_lens_forward " << names[1] << "\n";
    if (!names[2].empty()) src << "#define BK_HAS_GLOBE_PLATE 1\n#define LF_globe_plate " << names[2] << "\n";
    src << "#include \"bk_build_kernels.h\"\n";
    return src.str();
}

}  // namespace bk
