// bk_lua.cpp -- lexer, parser and host interpreter for the Lua 5.2 subset (see bk_lua.h).
#include "bk_lua.h"

#include <mutex>
#include <thread>
#include <condition_variable>

#include <cerrno>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "bkm.h"

namespace bklua {

// ---- math backends ---------------------------------------------------------------------------
static double p_sin(double x) { return bkm_sin(x); }
static double p_cos(double x) { return bkm_cos(x); }
static double p_tan(double x) { return bkm_tan(x); }
static double p_asin(double x) { return bkm_asin(x); }
static double p_acos(double x) { return bkm_acos(x); }
static double p_atan(double x) { return bkm_atan(x); }
static double p_atan2(double y, double x) { return bkm_atan2(y, x); }
static double p_sinh(double x) { return bkm_sinh(x); }
static double p_cosh(double x) { return bkm_cosh(x); }
static double p_tanh(double x) { return bkm_tanh(x); }
static double p_exp(double x) { return bkm_exp(x); }
static double p_log(double x) { return bkm_log(x); }
static double p_log10(double x) { return bkm_log10(x); }
static double p_pow(double x, double y) { return bkm_pow(x, y); }
static double p_sqrt(double x) { return bkm_sqrt(x); }
static double p_fmod(double x, double y) { return bkm_fmod(x, y); }

const MathLib &math_portable()
{
    static const MathLib m = {p_sin, p_cos, p_tan, p_asin, p_acos, p_atan, p_atan2, p_sinh, p_cosh, p_tanh,
                              p_exp, p_log, p_log10, p_pow, p_sqrt, p_fmod};
    return m;
}
const MathLib &math_platform()
{
    static const MathLib m = {::sin, ::cos, ::tan, ::asin, ::acos, ::atan, ::atan2, ::sinh, ::cosh, ::tanh,
                              ::exp, ::log, ::log10, ::pow, ::sqrt, ::fmod};
    return m;
}

// test-only: bkm.h with every inexact result moved by a pseudo-random relative amount of at most g_perturb_rel, i.e. "some
// other libm within that distance of bkm.h" (tests/test_exactness_cpu.py checks the device code's exactness flags against
// it).  Left alone: zeros / non-finite results, sqrt and fmod (correctly rounded / exact everywhere), atan2 on an axis
// (exact multiples of pi/2 in every libm, which bk_f_atan2 relies on).
static double g_perturb_rel = 0;
static int g_perturb_mode = 0;          // 0 pseudo-random, 1 always high, 2 always low
static double perturb(double r, double x, double y)
{
    if (r == 0 || !(r - r == 0)) return r;
    if (g_perturb_mode) return r + fabs(r) * ((g_perturb_mode == 1 ? 0.999 : -0.999) * g_perturb_rel);   // (the addition rounds)
    uint64_t a, b, h;
    memcpy(&a, &x, 8); memcpy(&b, &y, 8); memcpy(&h, &r, 8);
    h ^= a * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h ^= b * 0xC2B2AE3D27D4EB4Full; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const double u = (double)(h >> 11) * 0x1p-52 - 1.0;      // [-1, 1)
    return r + r * (u * g_perturb_rel);
}
#define BK_PERTURBED1(f) static double q_##f(double x) { return perturb(bkm_##f(x), x, 0); }
BK_PERTURBED1(sin) BK_PERTURBED1(cos) BK_PERTURBED1(tan) BK_PERTURBED1(asin) BK_PERTURBED1(acos) BK_PERTURBED1(atan)
BK_PERTURBED1(sinh) BK_PERTURBED1(cosh) BK_PERTURBED1(tanh) BK_PERTURBED1(exp) BK_PERTURBED1(log) BK_PERTURBED1(log10)
static double q_atan2(double y, double x) { return x == 0 || y == 0 ? bkm_atan2(y, x) : perturb(bkm_atan2(y, x), y, x); }
static double q_pow(double x, double y) { return perturb(bkm_pow(x, y), x, y); }
const MathLib &math_perturbed(double rel, int mode)
{
    g_perturb_mode = mode;
    static const MathLib m = {q_sin, q_cos, q_tan, q_asin, q_acos, q_atan, q_atan2, q_sinh, q_cosh, q_tanh,
                              q_exp, q_log, q_log10, q_pow, p_sqrt, p_fmod};
    g_perturb_rel = rel;
    return m;
}

// ---- lexer -------------------------------------------------------------------------------------
enum Tok {
    T_EOF, T_NAME, T_NUMBER, T_STRING,
    // keywords
    T_AND, T_BREAK, T_DO, T_ELSE, T_ELSEIF, T_END, T_FALSE, T_FOR, T_FUNCTION, T_GOTO, T_IF, T_IN, T_LOCAL,
    T_NIL, T_NOT, T_OR, T_REPEAT, T_RETURN, T_THEN, T_TRUE, T_UNTIL, T_WHILE,
    // multi-char operators
    T_EQ, T_NE, T_LE, T_GE, T_CONCAT, T_DOTS, T_DBCOLON,
    T_CHAR   // single character, in Token::ch
};

struct Token {
    Tok t = T_EOF;
    char ch = 0;
    double num = 0;
    std::string str;
    int line = 1;
};

static const struct { const char *w; Tok t; } KEYWORDS[] = {
    {"and", T_AND}, {"break", T_BREAK}, {"do", T_DO}, {"else", T_ELSE}, {"elseif", T_ELSEIF}, {"end", T_END},
    {"false", T_FALSE}, {"for", T_FOR}, {"function", T_FUNCTION}, {"goto", T_GOTO}, {"if", T_IF}, {"in", T_IN},
    {"local", T_LOCAL}, {"nil", T_NIL}, {"not", T_NOT}, {"or", T_OR}, {"repeat", T_REPEAT}, {"return", T_RETURN},
    {"then", T_THEN}, {"true", T_TRUE}, {"until", T_UNTIL}, {"while", T_WHILE}};

struct Lexer {
    const std::string &src;
    std::string chunk;
    size_t p = 0;
    int line = 1;
    Lexer(const std::string &s, const std::string &c) : src(s), chunk(c) {}

    [[noreturn]] void error(const std::string &msg, int ln) const
    {
        throw LuaError(chunk + ":" + std::to_string(ln) + ": " + msg);
    }
    int peekc(size_t o = 0) const { return p + o < src.size() ? (unsigned char)src[p + o] : -1; }

    // [[ ... ]] or [==[ ... ]==]; p at the first '['.  Returns false if this is not a long bracket.
    bool long_bracket(std::string *out)
    {
        size_t q = p + 1;
        int level = 0;
        while (q < src.size() && src[q] == '=') { ++level; ++q; }
        if (q >= src.size() || src[q] != '[') return false;
        ++q;
        if (q < src.size() && src[q] == '\n') { ++line; ++q; }        // first newline is skipped
        const int start_line = line;
        std::string close = "]" + std::string((size_t)level, '=') + "]";
        size_t e = src.find(close, q);
        if (e == std::string::npos) error("unfinished long string/comment", start_line);
        for (size_t i = q; i < e; ++i) if (src[i] == '\n') ++line;
        if (out) *out = src.substr(q, e - q);
        p = e + close.size();
        return true;
    }

    Token next()
    {
        for (;;) {
            int c = peekc();
            if (c == -1) { Token t; t.t = T_EOF; t.line = line; return t; }
            if (c == '\n') { ++line; ++p; continue; }
            if (c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v') { ++p; continue; }
            if (c == '-' && peekc(1) == '-') {
                p += 2;
                if (peekc() == '[' && long_bracket(nullptr)) continue;
                while (peekc() != -1 && peekc() != '\n') ++p;
                continue;
            }
            break;
        }
        Token t;
        t.line = line;
        int c = peekc();
        if (isalpha(c) || c == '_') {
            size_t s = p;
            while (isalnum(peekc()) || peekc() == '_') ++p;
            t.str = src.substr(s, p - s);
            t.t = T_NAME;
            for (auto &k : KEYWORDS) if (t.str == k.w) { t.t = k.t; break; }
            return t;
        }
        if (isdigit(c) || (c == '.' && isdigit(peekc(1)))) {
            // the stock lexer collects [0-9a-zA-Z.+-after-exponent] and hands it to strtod; so do we
            size_t s = p;
            bool hex = c == '0' && (peekc(1) == 'x' || peekc(1) == 'X');
            if (hex) p += 2;
            for (;;) {
                int d = peekc();
                if (d == -1) break;
                if ((hex ? (d == 'p' || d == 'P') : (d == 'e' || d == 'E')) && (peekc(1) == '+' || peekc(1) == '-')) { p += 2; continue; }
                if (isalnum(d) || d == '.') { ++p; continue; }
                break;
            }
            std::string lit = src.substr(s, p - s);
            char *end = nullptr;
            t.num = strtod(lit.c_str(), &end);
            if (!end || *end) error("malformed number near '" + lit + "'", line);
            t.t = T_NUMBER;
            return t;
        }
        if (c == '"' || c == '\'') {
            ++p;
            std::string s;
            for (;;) {
                int d = peekc();
                if (d == -1 || d == '\n') error("unfinished string", line);
                ++p;
                if (d == c) break;
                if (d == '\\') {
                    int e = peekc();
                    ++p;
                    switch (e) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'r': s += '\r'; break;
                    case 'a': s += '\a'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'v': s += '\v'; break;
                    case '\\': s += '\\'; break;
                    case '"': s += '"'; break;
                    case '\'': s += '\''; break;
                    case '\n': s += '\n'; ++line; break;
                    default:
                        if (isdigit(e)) {
                            int v = e - '0';
                            for (int k = 0; k < 2 && isdigit(peekc()); ++k) v = v * 10 + (src[p++] - '0');
                            s += (char)v;
                        } else error("invalid escape sequence", line);
                    }
                } else s += (char)d;
            }
            t.t = T_STRING;
            t.str = s;
            return t;
        }
        if (c == '[' && (peekc(1) == '[' || peekc(1) == '=')) {
            std::string s;
            if (long_bracket(&s)) { t.t = T_STRING; t.str = s; return t; }
        }
        ++p;
        auto two = [&](char second, Tok tk) { if (peekc() == second) { ++p; t.t = tk; return true; } return false; };
        switch (c) {
        case '=': if (two('=', T_EQ)) return t; break;
        case '~': if (two('=', T_NE)) return t; error("unexpected symbol near '~'", line);
        case '<': if (two('=', T_LE)) return t; break;
        case '>': if (two('=', T_GE)) return t; break;
        case ':': if (two(':', T_DBCOLON)) return t; break;
        case '.':
            if (peekc() == '.') { ++p; if (peekc() == '.') { ++p; t.t = T_DOTS; } else t.t = T_CONCAT; return t; }
            break;
        default: break;
        }
        t.t = T_CHAR;
        t.ch = (char)c;
        return t;
    }
};

// ---- parser ------------------------------------------------------------------------------------------
struct FuncState {
    FuncProto *f;
    FuncState *parent;
    std::vector<std::vector<std::pair<std::string, int>>> scopes;
};

struct Parser {
    Lexer lx;
    Token tok, ahead;
    bool has_ahead = false;
    std::shared_ptr<Chunk> chunk;
    FuncState *fs = nullptr;

    Parser(const std::string &src, const std::string &name) : lx(src, name)
    {
        chunk = std::make_shared<Chunk>();
        chunk->name = name;
        tok = lx.next();
    }
    [[noreturn]] void error(const std::string &msg) { lx.error(msg, tok.line); }
    void advance()
    {
        if (has_ahead) { tok = ahead; has_ahead = false; } else tok = lx.next();
    }
    const Token &lookahead()
    {
        if (!has_ahead) { ahead = lx.next(); has_ahead = true; }
        return ahead;
    }
    bool is_char(char c) const { return tok.t == T_CHAR && tok.ch == c; }
    bool accept_char(char c) { if (is_char(c)) { advance(); return true; } return false; }
    bool accept(Tok t) { if (tok.t == t) { advance(); return true; } return false; }
    void expect_char(char c, const char *what)
    {
        if (!accept_char(c)) error(std::string("'") + what + "' expected");
    }
    void expect(Tok t, const char *what) { if (!accept(t)) error(std::string("'") + what + "' expected"); }
    std::string expect_name()
    {
        if (tok.t != T_NAME) error("<name> expected");
        std::string s = tok.str;
        advance();
        return s;
    }

    // -- scopes
    void open_scope() { fs->scopes.emplace_back(); }
    void close_scope() { fs->scopes.pop_back(); }
    int declare_local(const std::string &name)
    {
        int slot = fs->f->nslots++;
        fs->f->slot_names.push_back(name);
        fs->scopes.back().emplace_back(name, slot);
        return slot;
    }
    static int find_local(FuncState *s, const std::string &name)
    {
        for (auto sc = s->scopes.rbegin(); sc != s->scopes.rend(); ++sc)
            for (auto v = sc->rbegin(); v != sc->rend(); ++v)
                if (v->first == name) return v->second;
        return -1;
    }
    static int find_upval(FuncState *s, const std::string &name)
    {
        for (size_t i = 0; i < s->f->upvals.size(); ++i)
            if (s->f->upvals[i].name == name) return (int)i;
        if (!s->parent) return -1;
        int l = find_local(s->parent, name);
        if (l >= 0) {
            if (s->parent->f->captured.size() <= (size_t)l) s->parent->f->captured.resize((size_t)l + 1, 0);
            s->parent->f->captured[(size_t)l] = 1;
            s->f->upvals.push_back({true, l, name});
            return (int)s->f->upvals.size() - 1;
        }
        int u = find_upval(s->parent, name);
        if (u < 0) return -1;
        s->f->upvals.push_back({false, u, name});
        return (int)s->f->upvals.size() - 1;
    }
    ExprP name_expr(const std::string &name, int line)
    {
        ExprP e(new Expr());
        e->kind = Expr::Name;
        e->line = line;
        e->str = name;
        int l = find_local(fs, name);
        if (l >= 0) { e->var = VarKind::Local; e->slot = l; return e; }
        int u = find_upval(fs, name);
        if (u >= 0) { e->var = VarKind::Upvalue; e->slot = u; return e; }
        e->var = VarKind::Global;
        e->slot = global_id(name);
        return e;
    }

    // -- functions
    FuncProto *new_proto(const std::string &name, int line)
    {
        chunk->protos.emplace_back(new FuncProto());
        FuncProto *f = chunk->protos.back().get();
        f->name = name;
        f->line = line;
        f->id = (int)chunk->protos.size() - 1;
        return f;
    }
    ExprP function_body(const std::string &name, int line, bool method = false)
    {
        FuncProto *f = new_proto(name, line);
        f->parent = fs ? fs->f : nullptr;
        FuncState nfs{f, fs, {}};
        fs = &nfs;
        open_scope();
        if (method) { declare_local("self"); f->nparams++; }      // function a:b(..) is a.b = function(self, ..)
        expect_char('(', "(");
        if (!is_char(')')) {
            do {
                if (tok.t == T_DOTS) { advance(); f->is_vararg = true; break; }
                declare_local(expect_name());
                f->nparams++;
            } while (accept_char(','));
        }
        expect_char(')', ")");
        f->body = block();
        expect(T_END, "end");
        close_scope();
        fs = nfs.parent;
        ExprP e(new Expr());
        e->kind = Expr::Function;
        e->line = line;
        e->proto = f;
        return e;
    }

    // -- expressions
    ExprP mk(Expr::Kind k)
    {
        ExprP e(new Expr());
        e->kind = k;
        e->line = tok.line;
        return e;
    }
    ExprP table_constructor()
    {
        ExprP t = mk(Expr::Table);
        expect_char('{', "{");
        while (!is_char('}')) {
            if (tok.t == T_NAME && lookahead().t == T_CHAR && lookahead().ch == '=') {
                ExprP k = mk(Expr::String);
                k->str = tok.str;
                advance(); advance();
                t->fields.emplace_back(std::move(k), expr());
                t->item_order.push_back(-(int)t->fields.size());
            } else if (is_char('[')) {
                advance();
                ExprP k = expr();
                expect_char(']', "]");
                expect_char('=', "=");
                t->fields.emplace_back(std::move(k), expr());
                t->item_order.push_back(-(int)t->fields.size());
            } else {
                t->args.push_back(expr());
                t->item_order.push_back((int)t->args.size() - 1);
            }
            if (!accept_char(',') && !accept_char(';')) break;
        }
        expect_char('}', "}");
        return t;
    }
    std::vector<ExprP> call_args()
    {
        std::vector<ExprP> args;
        if (tok.t == T_STRING) {
            ExprP s = mk(Expr::String);
            s->str = tok.str;
            advance();
            args.push_back(std::move(s));
            return args;
        }
        if (is_char('{')) { args.push_back(table_constructor()); return args; }
        expect_char('(', "(");
        if (!is_char(')')) {
            do args.push_back(expr()); while (accept_char(','));
        }
        expect_char(')', ")");
        return args;
    }
    ExprP primary_expr()
    {
        if (tok.t == T_NAME) {
            std::string n = tok.str;
            int line = tok.line;
            advance();
            return name_expr(n, line);
        }
        if (accept_char('(')) {
            ExprP e = expr();
            expect_char(')', ")");
            // parenthesised call/vararg is truncated to one value: wrap in a no-op unary
            if (e->kind == Expr::Call || e->kind == Expr::Vararg) {
                ExprP w = mk(Expr::Unop);
                w->str = "()";
                w->op = Expr::OP_PAREN;
                w->a = std::move(e);
                return w;
            }
            return e;
        }
        error("unexpected symbol");
    }
    ExprP suffixed_expr()
    {
        ExprP e = primary_expr();
        for (;;) {
            if (accept_char('.')) {
                ExprP k = mk(Expr::String);
                k->str = expect_name();
                ExprP i = mk(Expr::Index);
                i->a = std::move(e);
                i->b = std::move(k);
                e = std::move(i);
            } else if (is_char('[')) {
                advance();
                ExprP i = mk(Expr::Index);
                i->a = std::move(e);
                i->b = expr();
                expect_char(']', "]");
                e = std::move(i);
            } else if (is_char(':')) {                       // a:b(args): Call with the method's name in `str`; `a` is evaluated once
                advance();
                ExprP c = mk(Expr::Call);
                c->str = expect_name();
                c->a = std::move(e);
                c->args = call_args();
                e = std::move(c);
            } else if (is_char('(') || is_char('{') || tok.t == T_STRING) {
                ExprP c = mk(Expr::Call);
                c->a = std::move(e);
                c->args = call_args();
                e = std::move(c);
            } else {
                return e;
            }
        }
    }
    ExprP simple_expr()
    {
        ExprP e;
        switch (tok.t) {
        case T_NUMBER: e = mk(Expr::Number); e->num = tok.num; advance(); return e;
        case T_STRING: e = mk(Expr::String); e->str = tok.str; advance(); return e;
        case T_NIL: e = mk(Expr::Nil); advance(); return e;
        case T_TRUE: e = mk(Expr::True); advance(); return e;
        case T_FALSE: e = mk(Expr::False); advance(); return e;
        case T_DOTS:
            if (!fs->f->is_vararg) error("cannot use '...' outside a vararg function");
            e = mk(Expr::Vararg); advance(); return e;
        case T_FUNCTION: { int line = tok.line; advance(); return function_body("anonymous", line); }
        default:
            if (is_char('{')) return table_constructor();
            return suffixed_expr();
        }
    }
    struct OpInfo { const char *name; int left, right; };
    bool binop_info(OpInfo *o) const
    {
        switch (tok.t) {
        case T_OR: *o = {"or", 1, 1}; return true;
        case T_AND: *o = {"and", 2, 2}; return true;
        case T_EQ: *o = {"==", 3, 3}; return true;
        case T_NE: *o = {"~=", 3, 3}; return true;
        case T_LE: *o = {"<=", 3, 3}; return true;
        case T_GE: *o = {">=", 3, 3}; return true;
        case T_CONCAT: *o = {"..", 5, 4}; return true;
        case T_CHAR:
            switch (tok.ch) {
            case '<': *o = {"<", 3, 3}; return true;
            case '>': *o = {">", 3, 3}; return true;
            case '+': *o = {"+", 6, 6}; return true;
            case '-': *o = {"-", 6, 6}; return true;
            case '*': *o = {"*", 7, 7}; return true;
            case '/': *o = {"/", 7, 7}; return true;
            case '%': *o = {"%", 7, 7}; return true;
            case '^': *o = {"^", 10, 9}; return true;
            default: return false;
            }
        default: return false;
        }
    }
    ExprP subexpr(int limit)
    {
        ExprP e;
        const char *un = nullptr;
        if (tok.t == T_NOT) un = "not";
        else if (is_char('-')) un = "-";
        else if (is_char('#')) un = "#";
        if (un) {
            ExprP u = mk(Expr::Unop);
            u->str = un;
            u->op = un[0] == 'n' ? Expr::OP_NOT : un[0] == '-' ? Expr::OP_NEG : Expr::OP_LEN;
            advance();
            u->a = subexpr(8);                                   // UNARY_PRIORITY
            // fold -<number literal> like the stock compiler (exact)
            if (u->str == "-" && u->a->kind == Expr::Number) { u->a->num = -u->a->num; e = std::move(u->a); }
            else e = std::move(u);
        } else {
            e = simple_expr();
        }
        OpInfo op;
        while (binop_info(&op) && op.left > limit) {
            ExprP b = mk(Expr::Binop);
            b->str = op.name;
            {
                static const std::pair<const char *, Expr::Op> ops[] = {
                    {"+", Expr::OP_ADD}, {"-", Expr::OP_SUB}, {"*", Expr::OP_MUL}, {"/", Expr::OP_DIV}, {"%", Expr::OP_MOD}, {"^", Expr::OP_POW},
                    {"..", Expr::OP_CONCAT}, {"==", Expr::OP_EQ}, {"~=", Expr::OP_NE}, {"<", Expr::OP_LT}, {"<=", Expr::OP_LE}, {">", Expr::OP_GT},
                    {">=", Expr::OP_GE}, {"and", Expr::OP_AND}, {"or", Expr::OP_OR}};
                for (const auto &o : ops) if (b->str == o.first) b->op = o.second;
            }
            advance();
            b->a = std::move(e);
            b->b = subexpr(op.right);
            e = std::move(b);
        }
        return e;
    }
    ExprP expr() { return subexpr(0); }
    std::vector<ExprP> exprlist()
    {
        std::vector<ExprP> v;
        do v.push_back(expr()); while (accept_char(','));
        return v;
    }

    // -- statements
    bool block_follow() const
    {
        return tok.t == T_EOF || tok.t == T_END || tok.t == T_ELSE || tok.t == T_ELSEIF || tok.t == T_UNTIL;
    }
    StmtP mks(Stmt::Kind k)
    {
        StmtP s(new Stmt());
        s->kind = k;
        s->line = tok.line;
        return s;
    }
    Block block()
    {
        Block b;
        open_scope();
        while (!block_follow()) {
            if (tok.t == T_RETURN) {
                StmtP s = mks(Stmt::Return);
                advance();
                if (!block_follow() && !is_char(';')) s->exprs = exprlist();
                accept_char(';');
                b.push_back(std::move(s));
                break;
            }
            StmtP s = statement();
            if (s) b.push_back(std::move(s));
        }
        close_scope();
        return b;
    }
    StmtP statement()
    {
        switch (tok.t) {
        case T_CHAR:
            if (tok.ch == ';') { advance(); return nullptr; }
            break;
        case T_IF: {
            StmtP s = mks(Stmt::If);
            advance();
            ExprP c = expr();
            expect(T_THEN, "then");
            s->clauses.emplace_back(std::move(c), block());
            for (;;) {
                if (accept(T_ELSEIF)) {
                    ExprP c2 = expr();
                    expect(T_THEN, "then");
                    s->clauses.emplace_back(std::move(c2), block());
                } else if (accept(T_ELSE)) {
                    s->clauses.emplace_back(nullptr, block());
                    expect(T_END, "end");
                    break;
                } else { expect(T_END, "end"); break; }
            }
            return s;
        }
        case T_WHILE: {
            StmtP s = mks(Stmt::While);
            advance();
            s->cond = expr();
            expect(T_DO, "do");
            s->body = block();
            expect(T_END, "end");
            return s;
        }
        case T_DO: {
            StmtP s = mks(Stmt::Do);
            advance();
            s->body = block();
            expect(T_END, "end");
            return s;
        }
        case T_FOR: {
            int line = tok.line;
            advance();
            std::string n1 = expect_name();
            if (is_char('=')) {
                StmtP s = mks(Stmt::NumFor);
                s->line = line;
                advance();
                s->exprs.push_back(expr());
                expect_char(',', ",");
                s->exprs.push_back(expr());
                if (accept_char(',')) s->exprs.push_back(expr());
                expect(T_DO, "do");
                open_scope();
                s->slots.push_back(declare_local(n1));
                s->names.push_back(n1);
                s->body = block();
                close_scope();
                expect(T_END, "end");
                return s;
            }
            StmtP s = mks(Stmt::GenFor);
            s->line = line;
            s->names.push_back(n1);
            while (accept_char(',')) s->names.push_back(expect_name());
            expect(T_IN, "in");
            s->exprs = exprlist();
            expect(T_DO, "do");
            open_scope();
            for (auto &n : s->names) s->slots.push_back(declare_local(n));
            s->body = block();
            close_scope();
            expect(T_END, "end");
            return s;
        }
        case T_REPEAT: {
            StmtP s = mks(Stmt::Repeat);
            advance();
            // the until-condition sees the body's locals: parse the body without closing its scope
            open_scope();
            Block b;
            while (!block_follow()) {
                if (tok.t == T_RETURN) {
                    StmtP r = mks(Stmt::Return);
                    advance();
                    if (!block_follow() && !is_char(';')) r->exprs = exprlist();
                    accept_char(';');
                    b.push_back(std::move(r));
                    break;
                }
                StmtP st = statement();
                if (st) b.push_back(std::move(st));
            }
            expect(T_UNTIL, "until");
            s->cond = expr();
            close_scope();
            s->body = std::move(b);
            return s;
        }
        case T_FUNCTION: {
            int line = tok.line;
            advance();
            std::string n = expect_name();
            ExprP target = name_expr(n, line);
            std::string full = n;
            while (accept_char('.')) {
                ExprP k = mk(Expr::String);
                k->str = expect_name();
                full += "." + k->str;
                ExprP i = mk(Expr::Index);
                i->a = std::move(target);
                i->b = std::move(k);
                target = std::move(i);
            }
            bool method = false;
            if (accept_char(':')) {
                method = true;
                ExprP k = mk(Expr::String);
                k->str = expect_name();
                full += ":" + k->str;
                ExprP i = mk(Expr::Index);
                i->a = std::move(target);
                i->b = std::move(k);
                target = std::move(i);
            }
            StmtP s = mks(Stmt::Assign);
            s->line = line;
            s->targets.push_back(std::move(target));
            s->exprs.push_back(function_body(full, line, method));
            return s;
        }
        case T_LOCAL: {
            int line = tok.line;
            advance();
            if (accept(T_FUNCTION)) {
                StmtP s = mks(Stmt::LocalFunction);
                s->line = line;
                std::string n = expect_name();
                s->names.push_back(n);
                s->slots.push_back(declare_local(n));        // visible inside its own body (recursion)
                s->exprs.push_back(function_body(n, line));
                return s;
            }
            StmtP s = mks(Stmt::Local);
            s->line = line;
            do s->names.push_back(expect_name()); while (accept_char(','));
            if (accept_char('=')) s->exprs = exprlist();
            for (auto &n : s->names) s->slots.push_back(declare_local(n));   // after the initialisers
            return s;
        }
        case T_RETURN: error("'return' must be the last statement of a block");
        case T_BREAK: { StmtP s = mks(Stmt::Break); advance(); return s; }
        case T_GOTO: { StmtP s = mks(Stmt::Goto); advance(); s->names.push_back(expect_name()); return s; }
        case T_DBCOLON: {
            StmtP s = mks(Stmt::Label);
            advance();
            s->names.push_back(expect_name());
            if (tok.t != T_DBCOLON) error("'::' expected");
            advance();
            return s;
        }
        default: break;
        }
        // expression statement: call or assignment
        int line = tok.line;
        ExprP e = suffixed_expr();
        if (is_char('=') || is_char(',')) {
            StmtP s = mks(Stmt::Assign);
            s->line = line;
            s->targets.push_back(std::move(e));
            while (accept_char(',')) s->targets.push_back(suffixed_expr());
            expect_char('=', "=");
            s->exprs = exprlist();
            for (auto &t : s->targets)
                if (t->kind != Expr::Name && t->kind != Expr::Index) error("syntax error: cannot assign to this expression");
            return s;
        }
        if (e->kind != Expr::Call) error("syntax error near unexpected expression");
        StmtP s = mks(Stmt::CallStmt);
        s->line = line;
        s->call = std::move(e);
        return s;
    }

    std::shared_ptr<Chunk> parse_chunk()
    {
        FuncProto *m = new_proto("main chunk", 0);
        m->is_vararg = true;
        FuncState mfs{m, nullptr, {}};
        fs = &mfs;
        m->body = block();
        if (tok.t != T_EOF) error("'<eof>' expected");
        fs = nullptr;
        return chunk;
    }
};

std::shared_ptr<Chunk> parse(const std::string &src, const std::string &chunkname)
{
    Parser p(src, chunkname);
    return p.parse_chunk();
}

// ---- values ---------------------------------------------------------------------------------------------
const char *Value::type_name() const
{
    switch (t) {
    case NIL: return "nil";
    case BOOL: return "boolean";
    case NUM: return "number";
    case STR: return "string";
    case TABLE: return "table";
    case THREAD: return "thread";
    default: return "function";
    }
}

Value Table::get(const Value &k) const
{
    if (k.t == Value::NUM) {
        double d = k.n;
        if (d >= 1 && d <= (double)arr.size() && d == std::floor(d)) return arr[(size_t)d - 1];
        auto it = nhash.find(d);
        return it == nhash.end() ? Value() : it->second;
    }
    if (k.t == Value::STR) {
        auto it = shash.find(k.str());
        return it == shash.end() ? Value() : it->second;
    }
    return Value();
}

void Table::set(const Value &k, const Value &v)
{
    if (k.t == Value::NUM) {
        double d = k.n;
        if (d != d) throw LuaError("table index is NaN");
        if (d >= 1 && d == std::floor(d) && d <= (double)arr.size() + 1) {
            size_t i = (size_t)d;
            if (i <= arr.size()) {
                arr[i - 1] = v;
                if (v.t == Value::NIL && i == arr.size()) {               // shrink a trailing nil
                    while (!arr.empty() && arr.back().t == Value::NIL) arr.pop_back();
                }
                return;
            }
            if (v.t == Value::NIL) return;
            arr.push_back(v);
            for (;;) {                                                    // migrate following keys
                auto it = nhash.find((double)arr.size() + 1);
                if (it == nhash.end()) break;
                arr.push_back(it->second);
                nhash.erase(it);
            }
            return;
        }
        if (v.t == Value::NIL) nhash.erase(d); else nhash[d] = v;
        return;
    }
    if (k.t == Value::STR) {
        if (v.t == Value::NIL) shash.erase(k.str()); else shash[k.str()] = v;
        return;
    }
    if (k.t == Value::NIL) throw LuaError("table index is nil");
    throw LuaError(std::string("unsupported table key type: ") + k.type_name());
}

// ---- interpreter --------------------------------------------------------------------------------------------
namespace {

struct Frame {
    Closure *cl;
    bool any_captured = false;                     // (some local of this function lives in a shared cell)
    ValuesN<16> regs;                              // locals no inner function refers to
    std::vector<std::shared_ptr<Value>> cells;     // locals that closures capture (FuncProto::captured)
    Values varargs;
};

enum Flow { F_NORMAL, F_BREAK, F_RETURN, F_GOTO };

struct Exec {
    Interp &I;
    explicit Exec(Interp &i) : I(i) {}

    [[noreturn]] void error(int line, const std::string &chunk, const std::string &msg)
    {
        throw LuaError(chunk + ":" + std::to_string(line) + ": " + msg);
    }
    const std::string &chunk_of(Frame &f) { return f.cl->chunk->name; }

    static bool tonumber(const Value &v, double *out)
    {
        if (v.t == Value::NUM) { *out = v.n; return true; }
        if (v.t == Value::STR) {
            const char *s = v.str().c_str();
            char *end = nullptr;
            while (isspace((unsigned char)*s)) ++s;
            if (!*s) return false;
            double d = strtod(s, &end);
            while (end && isspace((unsigned char)*end)) ++end;
            if (!end || *end) return false;
            *out = d;
            return true;
        }
        return false;
    }

    // ---- metatables (lvm.c / ltm.c): only reached where a plain value would have produced nil or an error --------------------
    static Value metamethod(const Value &v, const char *event)
    {
        if (v.t == Value::TABLE && v.tab()->meta) return v.tab()->meta->get(Value::string(event));
        return Value();
    }
    static bool callable(const Value &v) { return v.is_function() || metamethod(v, "__call").is_function(); }
    // a binary metamethod of the first operand, else of the second; false if neither has one
    bool binary_meta(const Value &a, const Value &b, const char *event, Value *out)
    {
        Value h = metamethod(a, event);
        if (h.t == Value::NIL) h = metamethod(b, event);
        if (h.t == Value::NIL) return false;
        Values r = I.call(h, Values{a, b});
        *out = r.empty() ? Value() : r[0];
        return true;
    }
    void set_index(Frame &f, const Expr &target, const Value &o, const Value &k, const Value &v, int depth = 0)
    {
        if (o.t != Value::TABLE) error(target.line, chunk_of(f), std::string("attempt to index a ") + o.type_name() + " value");
        Table &t = *o.tab();
        if (t.meta && t.get(k).t == Value::NIL) {
            Value h = t.meta->get(Value::string("__newindex"));
            if (h.is_function()) { I.call(h, Values{o, k, v}); return; }
            if (h.t != Value::NIL) {
                if (depth > 100) error(target.line, chunk_of(f), "loop in settable");
                set_index(f, target, h, k, v, depth + 1);
                return;
            }
        }
        try { t.set(k, v); } catch (LuaError &err) { error(target.line, chunk_of(f), err.what()); }
    }

    Value arith(Frame &f, const Expr &e, const std::string &op, const Value &a, const Value &b)
    {
        double x, y;
        if (!tonumber(a, &x) || !tonumber(b, &y)) {
            const char *event = op[0] == '+' ? "__add" : op[0] == '-' ? "__sub" : op[0] == '*' ? "__mul" : op[0] == '/' ? "__div" : op[0] == '%' ? "__mod" : "__pow";
            Value out;
            if (binary_meta(a, b, event, &out)) return out;
        }
        if (!tonumber(a, &x)) error(e.line, chunk_of(f), std::string("attempt to perform arithmetic on a ") + a.type_name() + " value");
        if (!tonumber(b, &y)) error(e.line, chunk_of(f), std::string("attempt to perform arithmetic on a ") + b.type_name() + " value");
        switch (op[0]) {
        case '+': return Value::number(x + y);
        case '-': return Value::number(x - y);
        case '*': return Value::number(x * y);
        case '/': return Value::number(x / y);
        case '%': return Value::number(x - std::floor(x / y) * y);      // luai_nummod
        case '^': return Value::number(I.math->pow(x, y));               // luai_numpow = pow()
        }
        error(e.line, chunk_of(f), "bad arithmetic operator");
    }
    static bool raw_equal(const Value &a, const Value &b)
    {
        if (a.t != b.t) return false;
        switch (a.t) {
        case Value::NIL: return true;
        case Value::BOOL: return a.b == b.b;
        case Value::NUM: return a.n == b.n;
        case Value::STR: return a.str() == b.str();
        default: return a.p == b.p;            // tables, closures, builtins: identity
        }
    }
    bool less(Frame &f, const Expr &e, const Value &a, const Value &b, bool or_equal)
    {
        if (a.t == Value::NUM && b.t == Value::NUM) return or_equal ? a.n <= b.n : a.n < b.n;
        if (a.t == Value::STR && b.t == Value::STR) return or_equal ? a.str() <= b.str() : a.str() < b.str();
        Value out;
        if (binary_meta(a, b, or_equal ? "__le" : "__lt", &out)) return out.truthy();
        if (or_equal && binary_meta(b, a, "__lt", &out)) return !out.truthy();          // a <= b as not (b < a), lvm.c:luaV_lessequal
        error(e.line, chunk_of(f), std::string("attempt to compare ") + a.type_name() + " with " + b.type_name());
    }

    Value index(Frame &f, const Expr &e, const Value &obj, const Value &key)
    {
        if (obj.t == Value::TABLE) {
            Value v = obj.tab()->get(key);
            if (v.t != Value::NIL || !obj.tab()->meta) return v;
            Value h = obj.tab()->meta->get(Value::string("__index"));
            if (h.t == Value::NIL) return v;
            if (h.is_function()) { Values r = I.call(h, Values{obj, key}); return r.empty() ? Value() : r[0]; }
            if (++I.depth > 180) { --I.depth; error(e.line, chunk_of(f), "loop in gettable"); }
            Value out;
            try { out = index(f, e, h, key); } catch (...) { --I.depth; throw; }
            --I.depth;
            return out;
        }
        if (obj.t == Value::STR) {                               // strings index the string library (("x"):len(), s:sub(1, 2))
            Value lib = I.get_global("string");
            if (lib.t == Value::TABLE) return lib.tab()->get(key);
        }
        std::string what = e.a && e.a->kind == Expr::Name ? " (" + std::string(e.a->var == VarKind::Global ? "global" : "local") + " '" + e.a->str + "')" : "";
        error(e.line, chunk_of(f), std::string("attempt to index a ") + obj.type_name() + " value" + what);
    }

    Value &local_cell(Frame &f, int slot)
    {
        if (!f.any_captured || !f.cl->proto->is_captured(slot)) return f.regs[(size_t)slot];
        if (!f.cells[slot]) f.cells[slot] = std::make_shared<Value>();
        return *f.cells[slot];
    }
    // a NEW variable in `slot` (a `local`, a loop variable of this iteration): closures made earlier keep the old cell
    static void fresh_local(Frame &f, int slot, const Value &v)
    {
        if (f.any_captured && f.cl->proto->is_captured(slot)) f.cells[slot] = std::make_shared<Value>(v);
        else f.regs[(size_t)slot] = v;
    }

    Value eval(Frame &f, const Expr &e)
    {
        switch (e.kind) {
        case Expr::Nil: return Value();
        case Expr::True: return Value::boolean(true);
        case Expr::False: return Value::boolean(false);
        case Expr::Number: return Value::number(e.num);
        case Expr::String: return Value::string(e.str);
        case Expr::Vararg: return f.varargs.empty() ? Value() : f.varargs[0];
        case Expr::Name:
            if (e.var == VarKind::Local) return local_cell(f, e.slot);
            if (e.var == VarKind::Upvalue) return *f.cl->upvals[e.slot];
            return I.global_ref(e.slot, e.str);
        case Expr::Index: {
            Value o = eval(f, *e.a);
            Value k = eval(f, *e.b);
            return index(f, e, o, k);
        }
        case Expr::Call: return eval_call1(f, e);
        case Expr::Function: {
            auto cl = std::make_shared<Closure>();
            cl->proto = e.proto;
            cl->chunk = f.cl->chunk;
            for (const UpvalDesc &u : e.proto->upvals) {
                if (u.from_parent_local) {
                    if (!f.cells[u.index]) f.cells[u.index] = std::make_shared<Value>();
                    cl->upvals.push_back(f.cells[u.index]);
                } else cl->upvals.push_back(f.cl->upvals[u.index]);
            }
            return Value::closure(std::move(cl));
        }
        case Expr::Table: {
            Value v = Value::table(std::make_shared<Table>());
            v.tab()->arr.reserve(e.args.size());
            double next = 1;
            for (size_t oi = 0; oi < e.item_order.size(); ++oi) {
                int o = e.item_order[oi];
                if (o >= 0) {
                    bool last = oi + 1 == e.item_order.size();
                    const Expr &item = *e.args[o];
                    if (last && (item.kind == Expr::Call || item.kind == Expr::Vararg)) {
                        for (const Value &x : eval_multi(f, item)) { v.tab()->set(Value::number(next), x); next += 1; }
                    } else {
                        v.tab()->set(Value::number(next), eval(f, item));
                        next += 1;
                    }
                } else {
                    const auto &fld = e.fields[-1 - o];
                    Value k = eval(f, *fld.first);
                    v.tab()->set(k, eval(f, *fld.second));
                }
            }
            return v;
        }
        case Expr::Unop: {
            if (e.op == Expr::OP_PAREN) return eval(f, *e.a);
            Value a = eval(f, *e.a);
            if (e.op == Expr::OP_NOT) return Value::boolean(!a.truthy());
            if (e.op == Expr::OP_NEG) {
                double x;
                if (!tonumber(a, &x)) {
                    Value out;
                    if (binary_meta(a, a, "__unm", &out)) return out;
                    error(e.line, chunk_of(f), std::string("attempt to perform arithmetic on a ") + a.type_name() + " value");
                }
                return Value::number(-x);
            }
            if (e.op == Expr::OP_LEN) {
                if (a.t == Value::STR) return Value::number((double)a.str().size());
                if (a.t == Value::TABLE) {
                    Value h = metamethod(a, "__len");
                    if (h.t != Value::NIL) { Values r = I.call(h, Values{a}); return r.empty() ? Value() : r[0]; }
                    return Value::number((double)a.tab()->length());
                }
                error(e.line, chunk_of(f), std::string("attempt to get length of a ") + a.type_name() + " value");
            }
            error(e.line, chunk_of(f), "bad unary operator");
        }
        case Expr::Binop: {
            const std::string &op = e.str;
            if (e.op == Expr::OP_AND) { Value a = eval(f, *e.a); return a.truthy() ? eval(f, *e.b) : a; }
            if (e.op == Expr::OP_OR) { Value a = eval(f, *e.a); return a.truthy() ? a : eval(f, *e.b); }
            Value a = eval(f, *e.a);
            Value b = eval(f, *e.b);
            if (a.t == Value::NUM && b.t == Value::NUM) {              // the common case, without the generic helpers
                switch (e.op) {
                case Expr::OP_ADD: return Value::number(a.n + b.n);
                case Expr::OP_SUB: return Value::number(a.n - b.n);
                case Expr::OP_MUL: return Value::number(a.n * b.n);
                case Expr::OP_DIV: return Value::number(a.n / b.n);
                case Expr::OP_LT: return Value::boolean(a.n < b.n);
                case Expr::OP_LE: return Value::boolean(a.n <= b.n);
                case Expr::OP_GT: return Value::boolean(b.n < a.n);
                case Expr::OP_GE: return Value::boolean(b.n <= a.n);
                case Expr::OP_EQ: return Value::boolean(a.n == b.n);
                case Expr::OP_NE: return Value::boolean(!(a.n == b.n));
                default: break;
                }
            }
            if (e.op == Expr::OP_EQ || e.op == Expr::OP_NE) {
                bool eq = raw_equal(a, b);
                if (!eq && a.t == Value::TABLE && b.t == Value::TABLE) {            // two different tables with the same __eq handler
                    Value h = metamethod(a, "__eq");
                    if (h.t != Value::NIL && raw_equal(h, metamethod(b, "__eq"))) { Values r = I.call(h, Values{a, b}); eq = !r.empty() && r[0].truthy(); }
                }
                return Value::boolean(e.op == Expr::OP_EQ ? eq : !eq);
            }
            if (e.op == Expr::OP_LT) return Value::boolean(less(f, e, a, b, false));
            if (e.op == Expr::OP_LE) return Value::boolean(less(f, e, a, b, true));
            if (e.op == Expr::OP_GT) return Value::boolean(less(f, e, b, a, false));
            if (e.op == Expr::OP_GE) return Value::boolean(less(f, e, b, a, true));
            if (e.op == Expr::OP_CONCAT) {
                Value out;
                if (((a.t != Value::STR && a.t != Value::NUM) || (b.t != Value::STR && b.t != Value::NUM)) && binary_meta(a, b, "__concat", &out)) return out;
                if ((a.t != Value::STR && a.t != Value::NUM) || (b.t != Value::STR && b.t != Value::NUM))
                    error(e.line, chunk_of(f), std::string("attempt to concatenate a ") + (a.t != Value::STR && a.t != Value::NUM ? a : b).type_name() + " value");
                return Value::string(I.tostring(a) + I.tostring(b));
            }
            return arith(f, e, op, a, b);
        }
        }
        return Value();
    }

    Values eval_list(Frame &f, const std::vector<ExprP> &list)
    {
        Values out;
        for (size_t i = 0; i < list.size(); ++i) {
            const Expr &e = *list[i];
            if (i + 1 == list.size() && (e.kind == Expr::Call || e.kind == Expr::Vararg)) {
                Values m = eval_multi(f, e);
                out.append(m.begin(), m.end());
            } else out.push_back(eval(f, e));
        }
        return out;
    }

    [[noreturn]] void not_callable(Frame &f, const Expr &e, const Value &fn)
    {
        std::string what;
        if (e.a->kind == Expr::Name) what = std::string(" (") + (e.a->var == VarKind::Global ? "global" : "local") + " '" + e.a->str + "')";
        else if (e.a->kind == Expr::Index && e.a->b->kind == Expr::String) what = " (field '" + e.a->b->str + "')";
        error(e.line, chunk_of(f), std::string("attempt to call a ") + fn.type_name() + " value" + what);
    }
    // sin(x), sqrt(x) ...: a one-argument math builtin on a number needs no argument or result lists
    bool call_f1(Frame &f, const Expr &e, const Value &fn, Value *out, Values *args)
    {
        if (fn.t != Value::BUILTIN || !fn.bi()->f1 || e.args.size() != 1) return false;
        const Expr &ae = *e.args[0];
        if (ae.kind == Expr::Call || ae.kind == Expr::Vararg) return false;      // (may expand to no value at all: the generic path reports it)
        Value a = eval(f, ae);
        if (a.t == Value::NUM) { *out = Value::number((I.math->*(fn.bi()->f1))(a.n)); return true; }
        args->push_back(std::move(a));                                            // evaluated once: the generic path takes it from here
        return false;
    }
    // a script function called with plain arguments: they are evaluated straight into the callee's frame
    bool call_direct(Frame &f, const Expr &e, const Value &fn, Values *rets)
    {
        if (fn.t != Value::FUNC) return false;
        Closure *cl = fn.fn();
        const FuncProto *p = cl->proto;
        if (p->is_vararg || !p->captured.empty()) return false;
        if (!e.args.empty()) { const Expr &last = *e.args.back(); if (last.kind == Expr::Call || last.kind == Expr::Vararg) return false; }
        if (I.depth > 180) throw LuaError("stack overflow (recursion too deep)");
        Frame fr;
        fr.cl = cl;
        fr.regs.resize((size_t)p->nslots);
        for (size_t i = 0; i < e.args.size(); ++i) {
            Value v = eval(f, *e.args[i]);                     // (arguments beyond the parameters are evaluated and dropped)
            if ((int)i < p->nparams) fr.regs[i] = std::move(v);
        }
        ++I.depth;
        try { if (exec_block(fr, p->body, *rets) == F_GOTO) throw LuaError(chunk_of(fr) + ": no visible label '" + I.goto_label + "' for goto"); } catch (...) { --I.depth; throw; }
        --I.depth;
        return true;
    }
    void note_call_site(Frame &f, const Expr &e) { I.call_chunk = &chunk_of(f); I.call_line = e.line; }
    // a:b(args): the function is a.b, its first argument a
    Values method_call(Frame &f, const Expr &e)
    {
        Value obj = eval(f, *e.a);
        Value fn = index(f, e, obj, Value::string(e.str));
        if (!callable(fn)) error(e.line, chunk_of(f), std::string("attempt to call a ") + fn.type_name() + " value (method '" + e.str + "')");
        Values args;
        args.push_back(std::move(obj));
        Values rest = eval_list(f, e.args);
        args.append(rest.begin(), rest.end());
        if (fn.t == Value::BUILTIN) note_call_site(f, e);
        return I.call(fn, args);
    }
    // a call of which only the first result is wanted
    Value eval_call1(Frame &f, const Expr &e)
    {
        if (!e.str.empty()) { Values r = method_call(f, e); return r.empty() ? Value() : std::move(r[0]); }
        Value fn = eval(f, *e.a);
        Value out;
        Values args;
        if (call_f1(f, e, fn, &out, &args)) return out;
        if (args.empty()) {
            Values r;
            if (call_direct(f, e, fn, &r)) return r.empty() ? Value() : std::move(r[0]);
        }
        if (args.empty()) args = eval_list(f, e.args);
        if (!callable(fn)) not_callable(f, e, fn);
        if (fn.t == Value::BUILTIN) note_call_site(f, e);
        Values r = I.call(fn, args);
        return r.empty() ? Value() : std::move(r[0]);
    }

    Values eval_multi(Frame &f, const Expr &e)
    {
        if (e.kind == Expr::Vararg) return f.varargs;
        if (e.kind != Expr::Call) return Values{eval(f, e)};
        if (!e.str.empty()) return method_call(f, e);
        Value fn = eval(f, *e.a);
        Value out;
        Values args;
        if (call_f1(f, e, fn, &out, &args)) return Values{out};
        if (args.empty()) {
            Values r;
            if (call_direct(f, e, fn, &r)) return r;
        }
        if (args.empty()) args = eval_list(f, e.args);
        if (!callable(fn)) not_callable(f, e, fn);
        if (fn.t == Value::BUILTIN) note_call_site(f, e);
        return I.call(fn, args);
    }

    void assign(Frame &f, const Expr &target, const Value &v)
    {
        if (target.kind == Expr::Name) {
            if (target.var == VarKind::Local) local_cell(f, target.slot) = v;
            else if (target.var == VarKind::Upvalue) *f.cl->upvals[target.slot] = v;
            else I.global_ref(target.slot, target.str) = v;
            return;
        }
        Value o = eval(f, *target.a);
        Value k = eval(f, *target.b);
        set_index(f, target, o, k, v);
    }

    void tick(Frame &f, int line)
    {
        if (++I.steps > I.max_steps) error(line, chunk_of(f), "script exceeded the execution budget (infinite loop?)");
    }

    Flow exec_block(Frame &f, const Block &b, Values &ret)
    {
        for (size_t i = 0; i < b.size();) {
            Flow fl = exec(f, *b[i], ret);
            if (fl == F_GOTO) {
                // a goto continues after its label in the nearest enclosing block that has it (checked as the jump happens, where the
                // reference's compiler checks while parsing); loops and blocks in between are left
                size_t at = b.size();
                for (size_t k = 0; k < b.size(); ++k)
                    if (b[k]->kind == Stmt::Label && b[k]->names[0] == I.goto_label) { at = k; break; }
                if (at == b.size()) return F_GOTO;
                tick(f, b[at]->line);
                i = at + 1;
                continue;
            }
            if (fl != F_NORMAL) return fl;
            ++i;
        }
        return F_NORMAL;
    }

    Flow exec(Frame &f, const Stmt &s, Values &ret)
    {
        tick(f, s.line);
        switch (s.kind) {
        case Stmt::Local: {
            if (s.slots.size() == 1 && s.exprs.size() == 1) { fresh_local(f, s.slots[0], eval(f, *s.exprs[0])); return F_NORMAL; }
            Values v = eval_list(f, s.exprs);
            for (size_t i = 0; i < s.slots.size(); ++i)
                fresh_local(f, s.slots[i], i < v.size() ? v[i] : Value());
            return F_NORMAL;
        }
        case Stmt::LocalFunction: {
            fresh_local(f, s.slots[0], Value());                       // (the function sees itself: declared before it is evaluated)
            local_cell(f, s.slots[0]) = eval(f, *s.exprs[0]);
            return F_NORMAL;
        }
        case Stmt::Assign: {
            if (s.targets.size() == 1 && s.exprs.size() == 1) { assign(f, *s.targets[0], eval(f, *s.exprs[0])); return F_NORMAL; }
            Values v = eval_list(f, s.exprs);
            for (size_t i = 0; i < s.targets.size(); ++i) assign(f, *s.targets[i], i < v.size() ? v[i] : Value());
            return F_NORMAL;
        }
        case Stmt::CallStmt: eval_multi(f, *s.call); return F_NORMAL;
        case Stmt::Do: return exec_block(f, s.body, ret);
        case Stmt::While:
            while (eval(f, *s.cond).truthy()) {
                tick(f, s.line);
                Flow fl = exec_block(f, s.body, ret);
                if (fl == F_BREAK) break;
                if (fl == F_RETURN || fl == F_GOTO) return fl;
            }
            return F_NORMAL;
        case Stmt::Repeat:
            for (;;) {
                tick(f, s.line);
                Flow fl = exec_block(f, s.body, ret);
                if (fl == F_BREAK) break;
                if (fl == F_RETURN || fl == F_GOTO) return fl;
                if (eval(f, *s.cond).truthy()) break;
            }
            return F_NORMAL;
        case Stmt::If:
            for (const auto &c : s.clauses)
                if (!c.first || eval(f, *c.first).truthy()) return exec_block(f, c.second, ret);
            return F_NORMAL;
        case Stmt::NumFor: {
            double start, stop, step = 1;
            if (!tonumber(eval(f, *s.exprs[0]), &start)) error(s.line, chunk_of(f), "'for' initial value must be a number");
            if (!tonumber(eval(f, *s.exprs[1]), &stop)) error(s.line, chunk_of(f), "'for' limit must be a number");
            if (s.exprs.size() > 2 && !tonumber(eval(f, *s.exprs[2]), &step)) error(s.line, chunk_of(f), "'for' step must be a number");
            // OP_FORPREP / OP_FORLOOP: idx = start - step; loop { idx += step; test; body }
            double idx = start - step;
            for (;;) {
                idx = idx + step;
                if (!(0 < step ? idx <= stop : stop <= idx)) break;
                tick(f, s.line);
                fresh_local(f, s.slots[0], Value::number(idx));
                Flow fl = exec_block(f, s.body, ret);
                if (fl == F_BREAK) break;
                if (fl == F_RETURN || fl == F_GOTO) return fl;
            }
            return F_NORMAL;
        }
        case Stmt::GenFor: {
            Values init = eval_list(f, s.exprs);
            init.resize(3);
            Value fn = init[0], st = init[1], ctl = init[2];
            for (;;) {
                tick(f, s.line);
                if (!callable(fn)) error(s.line, chunk_of(f), std::string("attempt to call a ") + fn.type_name() + " value");
                Values r = I.call(fn, Values{st, ctl});
                if (r.empty() || r[0].t == Value::NIL) break;
                ctl = r[0];
                for (size_t i = 0; i < s.slots.size(); ++i)
                    fresh_local(f, s.slots[i], i < r.size() ? r[i] : Value());
                Flow fl = exec_block(f, s.body, ret);
                if (fl == F_BREAK) break;
                if (fl == F_RETURN || fl == F_GOTO) return fl;
            }
            return F_NORMAL;
        }
        case Stmt::Return: ret = eval_list(f, s.exprs); return F_RETURN;
        case Stmt::Break: return F_BREAK;
        case Stmt::Goto: I.goto_label = s.names[0]; return F_GOTO;
        case Stmt::Label: return F_NORMAL;
        }
        return F_NORMAL;
    }
};

}  // namespace

int global_id(const std::string &name)
{
    static std::mutex m;
    static std::map<std::string, int> ids;
    std::lock_guard<std::mutex> lock(m);
    auto it = ids.find(name);
    if (it == ids.end()) it = ids.emplace(name, (int)ids.size()).first;
    return it->second;
}

Value Interp::get_global(const std::string &name) const
{
    auto it = globals.find(name);
    return it == globals.end() ? Value() : it->second;
}
void Interp::set_global(const std::string &name, const Value &v)
{
    ++activity;
    globals[name] = v;                        // (nil stays as a nil-valued node: gslots point at the nodes)
}
void Interp::register_builtin(const std::string &name, BuiltinFn fn)
{
    Value v;
    v.t = Value::BUILTIN;
    auto bp = std::make_shared<Builtin>();
    bp->name = name;
    bp->fn = std::move(fn);
    v.p = std::move(bp);
    size_t dot = name.find('.');
    if (dot == std::string::npos) { globals[name] = v; return; }
    std::string tname = name.substr(0, dot), field = name.substr(dot + 1);
    Value t = get_global(tname);
    if (t.t != Value::TABLE) { t = Value::table(std::make_shared<Table>()); globals[tname] = t; }
    t.tab()->shash[field] = v;
}

namespace {
struct Cloner {
    std::map<const Table *, std::shared_ptr<Table>> tabs;
    std::map<const Closure *, std::shared_ptr<Closure>> fns;
    std::map<const Value *, std::shared_ptr<Value>> cells;
    std::map<const Builtin *, std::shared_ptr<Builtin>> bis;
    Value value(const Value &v)
    {
        Value o = v;
        if (v.t == Value::TABLE) o.p = table(v.tab_ptr());
        else if (v.t == Value::FUNC) o.p = closure(v.fn_ptr());
        else if (v.t == Value::BUILTIN && v.p) {
            // (immutable, but every evaluation of `sin` copies the Value: a private copy keeps the reference count -
            // an atomic - out of the other threads' cache lines)
            auto it = bis.find(v.bi());
            if (it == bis.end()) it = bis.emplace(v.bi(), std::make_shared<Builtin>(*v.bi())).first;
            o.p = it->second;
        } else if (v.t == Value::STR && v.p) o.p = std::make_shared<std::string>(v.str());
        else if (v.t == Value::THREAD) o = Value();          // (a coroutine is a native stack of THIS interpreter: a copy of the state for another thread has none of it)
        return o;
    }
    std::shared_ptr<Table> table(const std::shared_ptr<Table> &t)
    {
        if (!t) return t;
        auto it = tabs.find(t.get());
        if (it != tabs.end()) return it->second;
        auto n = std::make_shared<Table>();
        tabs[t.get()] = n;                                   // (registered first: tables may refer to themselves)
        for (const Value &x : t->arr) n->arr.push_back(value(x));
        for (const auto &kv : t->nhash) n->nhash[kv.first] = value(kv.second);
        for (const auto &kv : t->shash) n->shash[kv.first] = value(kv.second);
        n->meta = table(t->meta);
        return n;
    }
    std::shared_ptr<Closure> closure(const std::shared_ptr<Closure> &c)
    {
        if (!c) return c;
        auto it = fns.find(c.get());
        if (it != fns.end()) return it->second;
        auto n = std::make_shared<Closure>();
        fns[c.get()] = n;
        n->proto = c->proto;
        n->chunk = c->chunk;
        for (const std::shared_ptr<Value> &u : c->upvals) {
            auto ci = cells.find(u.get());
            if (ci != cells.end()) { n->upvals.push_back(ci->second); continue; }
            auto cell = std::make_shared<Value>();
            cells[u.get()] = cell;
            n->upvals.push_back(cell);
            if (u) *cell = value(*u);
        }
        return n;
    }
};
}  // namespace

std::unique_ptr<Interp> Interp::clone(const Values &roots, Values *roots_out) const
{
    std::unique_ptr<Interp> n(new Interp(*math, Empty{}));
    Cloner c;
    for (const auto &kv : globals) n->globals[kv.first] = c.value(kv.second);
    n->max_steps = max_steps;
    n->print_sink = nullptr;
    if (roots_out) {
        roots_out->clear();
        for (const Value &r : roots) roots_out->push_back(c.value(r));
    }
    return n;
}

std::string Interp::tostring(const Value &v) const
{
    char buf[64];
    switch (v.t) {
    case Value::NIL: return "nil";
    case Value::BOOL: return v.b ? "true" : "false";
    case Value::NUM: snprintf(buf, sizeof buf, "%.14g", v.n); return buf;     // LUA_NUMBER_FMT
    case Value::STR: return v.str();
    case Value::TABLE:
        if (v.tab()->meta) {
            Value h = v.tab()->meta->get(Value::string("__tostring"));
            if (h.is_function()) {
                Values r = const_cast<Interp *>(this)->call(h, Values{v});
                if (r.empty() || r[0].t != Value::STR) throw LuaError("'__tostring' must return a string");
                return r[0].str();
            }
        }
        snprintf(buf, sizeof buf, "table: %p", (void *)v.tab());
        return buf;
    case Value::FUNC: snprintf(buf, sizeof buf, "function: %p", (void *)v.fn()); return buf;
    case Value::THREAD: snprintf(buf, sizeof buf, "thread: %p", v.p.get()); return buf;
    default: return "function: builtin: " + v.bi()->name;
    }
}

Values Interp::call(const Value &fv, const Values &args)
{
    ++activity;
    if (depth == 0) call_chunk = nullptr;       // (a call from the host: no script call site yet, and the last one's chunk may be gone)
    if (fv.t == Value::BUILTIN) {
        Values rets;
        // (a builtin that runs into a C++ exception of its own - std::length_error from a string of 1e300 bytes, bad_alloc - is a
        //  script error like any other: nothing but LuaError may unwind towards the C ABI, where the entry points catch exactly that)
        try {
            fv.bi()->fn(*this, args, rets);
        } catch (const LuaError &) {
            throw;
        } catch (const std::exception &e) {
            throw LuaError(fv.bi()->name + ": " + e.what());
        }
        return rets;
    }
    if (fv.t == Value::TABLE && fv.tab()->meta) {                   // __call: the table itself becomes the first argument
        Value h = fv.tab()->meta->get(Value::string("__call"));
        if (h.is_function()) {
            Values with_self;
            with_self.push_back(fv);
            with_self.append(args.begin(), args.end());
            return call(h, with_self);
        }
    }
    if (fv.t != Value::FUNC) throw LuaError(std::string("attempt to call a ") + fv.type_name() + " value");
    if (depth > 180) throw LuaError("stack overflow (recursion too deep)");
    const FuncProto *p = fv.fn()->proto;
    Frame fr;
    fr.cl = fv.fn();
    fr.regs.resize((size_t)p->nslots);
    if (!p->captured.empty()) {
        fr.cells.resize((size_t)p->nslots);
        for (char c : p->captured) fr.any_captured = fr.any_captured || c;
    }
    for (int i = 0; i < p->nparams; ++i) {
        if (p->is_captured(i)) fr.cells[i] = std::make_shared<Value>((size_t)i < args.size() ? args[i] : Value());
        else if ((size_t)i < args.size()) fr.regs[(size_t)i] = args[i];
    }
    if (p->is_vararg && args.size() > (size_t)p->nparams) fr.varargs.assign(args.begin() + p->nparams, args.end());
    Values ret;
    Exec ex(*this);
    ++depth;
    try {
        if (ex.exec_block(fr, p->body, ret) == F_GOTO) throw LuaError(fr.cl->chunk->name + ": no visible label '" + goto_label + "' for goto");
    } catch (...) { --depth; throw; }
    --depth;
    return ret;
}

void Interp::run(const std::string &src, const std::string &chunkname)
{
    ++activity;
    std::shared_ptr<Chunk> ch = parse(src, chunkname);
    Value f;
    f.t = Value::FUNC;
    auto cl = std::make_shared<Closure>();
    cl->proto = ch->main();
    cl->chunk = ch;
    f.p = std::move(cl);
    call(f, Values());
}


// ---- Lua patterns (manual 6.4.1): character classes, sets, the four quantifiers, anchors, captures and position captures,
// %b, %f and back-references.  A recursive matcher over (subject position, pattern position), written from the manual's rules.
namespace {
struct PatternMatch {
    const std::string &s, &p;
    struct Cap { size_t start; long len; };          // len -1: still open, -2: position capture
    std::vector<Cap> caps;
    int depth = 0;
    PatternMatch(const std::string &subject, const std::string &pattern) : s(subject), p(pattern) {}

    [[noreturn]] static void bad(const char *what) { throw LuaError(std::string("malformed pattern (") + what + ")"); }
    static bool class_matches(unsigned char c, unsigned char cl)
    {
        bool r;
        switch (tolower(cl)) {
        case 'a': r = isalpha(c) != 0; break;
        case 'c': r = iscntrl(c) != 0; break;
        case 'd': r = isdigit(c) != 0; break;
        case 'g': r = isgraph(c) != 0; break;
        case 'l': r = islower(c) != 0; break;
        case 'p': r = ispunct(c) != 0; break;
        case 's': r = isspace(c) != 0; break;
        case 'u': r = isupper(c) != 0; break;
        case 'w': r = isalnum(c) != 0; break;
        case 'x': r = isxdigit(c) != 0; break;
        default: return cl == c;                     // %<punctuation>: that character itself
        }
        return isupper(cl) ? !r : r;
    }
    // end of the single-character item that starts at pattern position pi
    size_t item_end(size_t pi) const
    {
        if (pi >= p.size()) bad("ends unexpectedly");
        const char c = p[pi++];
        if (c == '%') {
            if (pi >= p.size()) bad("ends with '%'");
            return pi + 1;
        }
        if (c == '[') {
            if (pi < p.size() && p[pi] == '^') ++pi;
            for (bool first = true;; first = false) {      // the first character after [ or [^ belongs to the set, even a ']'
                if (pi >= p.size()) bad("missing ']'");
                const char k = p[pi++];
                if (k == ']' && !first) return pi;
                if (k == '%') { if (pi >= p.size()) bad("missing ']'"); ++pi; }
            }
        }
        return pi;
    }
    bool set_matches(unsigned char c, size_t pi, size_t end) const        // p[pi] == '[', p[end - 1] == ']'
    {
        bool negate = false;
        ++pi;
        if (p[pi] == '^') { negate = true; ++pi; }
        const size_t last = end - 1;
        while (pi < last) {
            if (p[pi] == '%' && pi + 1 < last) { if (class_matches(c, (unsigned char)p[pi + 1])) return !negate; pi += 2; }
            else if (pi + 2 < last && p[pi + 1] == '-') { if ((unsigned char)p[pi] <= c && c <= (unsigned char)p[pi + 2]) return !negate; pi += 3; }
            else { if ((unsigned char)p[pi] == c) return !negate; ++pi; }
        }
        return negate;
    }
    bool single_matches(size_t si, size_t pi, size_t end) const
    {
        if (si >= s.size()) return false;
        const unsigned char c = (unsigned char)s[si];
        switch (p[pi]) {
        case '.': return true;
        case '%': return class_matches(c, (unsigned char)p[pi + 1]);
        case '[': return set_matches(c, pi, end);
        default: return (unsigned char)p[pi] == c;
        }
    }
    static const size_t NO = (size_t)-1;
    // the subject position after a match of p[pi..] at s[si..], or NO
    size_t match(size_t si, size_t pi)
    {
        if (++depth > 200) { --depth; throw LuaError("pattern too complex"); }
        struct Leave { int &d; ~Leave() { --d; } } leave{depth};
        for (;;) {
            if (pi >= p.size()) return si;
            switch (p[pi]) {
            case '(': {
                const bool position = pi + 1 < p.size() && p[pi + 1] == ')';
                caps.push_back({si, position ? -2 : -1});
                const size_t r = match(si, pi + (position ? 2 : 1));
                if (r == NO) caps.pop_back();
                return r;
            }
            case ')': {
                int open = -1;
                for (int i = (int)caps.size() - 1; i >= 0; --i) if (caps[(size_t)i].len == -1) { open = i; break; }
                if (open < 0) bad("invalid pattern capture");
                caps[(size_t)open].len = (long)(si - caps[(size_t)open].start);
                const size_t r = match(si, pi + 1);
                if (r == NO) caps[(size_t)open].len = -1;
                return r;
            }
            case '$':
                if (pi + 1 == p.size()) return si == s.size() ? si : NO;
                break;
            case '%':
                if (pi + 1 < p.size() && p[pi + 1] == 'b') {                       // %bxy: balanced
                    if (pi + 3 >= p.size()) bad("missing arguments to '%b'");
                    if (si >= s.size() || s[si] != p[pi + 2]) return NO;
                    const char open = p[pi + 2], close = p[pi + 3];
                    int level = 1;
                    size_t k = si + 1;
                    for (; k < s.size(); ++k) {
                        if (s[k] == close) { if (--level == 0) break; }
                        else if (s[k] == open) ++level;
                    }
                    if (k >= s.size()) return NO;
                    si = k + 1;
                    pi += 4;
                    continue;
                }
                if (pi + 1 < p.size() && p[pi + 1] == 'f') {                       // %f[set]: frontier
                    pi += 2;
                    if (pi >= p.size() || p[pi] != '[') bad("missing '[' after '%f' in pattern");
                    const size_t end = item_end(pi);
                    const unsigned char before = si == 0 ? 0 : (unsigned char)s[si - 1], here = si < s.size() ? (unsigned char)s[si] : 0;
                    if (set_matches(before, pi, end) || !set_matches(here, pi, end)) return NO;
                    pi = end;
                    continue;
                }
                if (pi + 1 < p.size() && isdigit((unsigned char)p[pi + 1])) {      // %1..%9: what that capture matched
                    const int idx = p[pi + 1] - '1';
                    if (idx < 0 || idx >= (int)caps.size() || caps[(size_t)idx].len < 0) bad("invalid capture index");
                    const size_t len = (size_t)caps[(size_t)idx].len;
                    if (s.size() - si < len || s.compare(si, len, s, caps[(size_t)idx].start, len) != 0) return NO;
                    si += len;
                    pi += 2;
                    continue;
                }
                break;
            default: break;
            }
            const size_t end = item_end(pi);
            const char q = end < p.size() ? p[end] : '\0';
            if (q == '?') {
                if (single_matches(si, pi, end)) { const size_t r = match(si + 1, end + 1); if (r != NO) return r; }
                pi = end + 1;
                continue;
            }
            if (q == '+' || q == '*') {                                            // longest first
                size_t n = 0;
                while (single_matches(si + n, pi, end)) ++n;
                if (q == '+' && n == 0) return NO;
                for (size_t k = n + 1; k-- > (q == '+' ? 1u : 0u);) { const size_t r = match(si + k, end + 1); if (r != NO) return r; }
                return NO;
            }
            if (q == '-') {                                                        // shortest first
                for (;;) {
                    const size_t r = match(si, end + 1);
                    if (r != NO) return r;
                    if (!single_matches(si, pi, end)) return NO;
                    ++si;
                }
            }
            if (!single_matches(si, pi, end)) return NO;
            ++si;
            pi = end;
        }
    }
    // capture i of a finished match as a value (the whole match if the pattern has no captures and i == 0)
    Value capture(size_t i, size_t start, size_t end) const
    {
        if (i >= caps.size()) {
            if (i == 0) return Value::string(s.substr(start, end - start));
            throw LuaError("invalid capture index");
        }
        if (caps[i].len == -2) return Value::number((double)caps[i].start + 1);
        if (caps[i].len < 0) throw LuaError("unfinished capture");
        return Value::string(s.substr(caps[i].start, (size_t)caps[i].len));
    }
    void push_captures(Values &r, size_t start, size_t end, bool whole_if_none) const
    {
        const size_t n = caps.empty() && whole_if_none ? 1 : caps.size();
        for (size_t i = 0; i < n; ++i) r.push_back(capture(i, start, end));
    }
};

// first match of `pat` in `subject` at or after `init` (0-based): (start, end) or false; `m` keeps the captures
bool pattern_search(PatternMatch &m, size_t init, size_t *start, size_t *end)
{
    const bool anchored = !m.p.empty() && m.p[0] == '^';
    for (size_t si = init; si <= m.s.size(); ++si) {
        m.caps.clear();
        const size_t e = m.match(si, anchored ? 1 : 0);
        if (e != PatternMatch::NO) { *start = si; *end = e; return true; }
        if (anchored) break;
    }
    return false;
}
}  // namespace

// ---- standard library subset ----------------------------------------------------------------------------
static double argnum(const Values &a, size_t i, const char *fn)
{
    double d;
    if (i >= a.size() || !Exec::tonumber(a[i], &d))
        throw LuaError(std::string("bad argument #") + std::to_string(i + 1) + " to '" + fn + "' (number expected, got " +
                       (i < a.size() ? a[i].type_name() : "no value") + ")");
    return d;
}

Interp::Interp(const MathLib &m) : math(&m)
{
    auto m1 = [this](const char *name, double (*MathLib::*fp)(double)) {
        std::string n = std::string("math.") + name;
        register_builtin(n, [this, fp, n](Interp &, const Values &a, Values &r) {
            r.push_back(Value::number((math->*fp)(argnum(a, 0, n.c_str() + 5))));
        });
        get_global("math").tab()->get(Value::string(name)).bi()->f1 = fp;       // (the interpreter's shortcut for a number argument)
    };
    m1("sin", &MathLib::sin); m1("cos", &MathLib::cos); m1("tan", &MathLib::tan);
    m1("asin", &MathLib::asin); m1("acos", &MathLib::acos); m1("atan", &MathLib::atan);
    m1("sinh", &MathLib::sinh); m1("cosh", &MathLib::cosh); m1("tanh", &MathLib::tanh);
    m1("exp", &MathLib::exp); m1("log10", &MathLib::log10); m1("sqrt", &MathLib::sqrt);
    register_builtin("math.atan2", [this](Interp &, const Values &a, Values &r) { r.push_back(Value::number(math->atan2(argnum(a, 0, "atan2"), argnum(a, 1, "atan2")))); });
    register_builtin("math.pow", [this](Interp &, const Values &a, Values &r) { r.push_back(Value::number(math->pow(argnum(a, 0, "pow"), argnum(a, 1, "pow")))); });
    register_builtin("math.fmod", [this](Interp &, const Values &a, Values &r) { r.push_back(Value::number(math->fmod(argnum(a, 0, "fmod"), argnum(a, 1, "fmod")))); });
    register_builtin("math.log", [this](Interp &, const Values &a, Values &r) {      // math_log, Lua 5.2
        double x = argnum(a, 0, "log");
        if (a.size() < 2 || a[1].t == Value::NIL) { r.push_back(Value::number(math->log(x))); return; }
        double b = argnum(a, 1, "log");
        r.push_back(Value::number(b == 10.0 ? math->log10(x) : math->log(x) / math->log(b)));
    });
    register_builtin("math.abs", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(std::fabs(argnum(a, 0, "abs")))); });
    register_builtin("math.floor", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(std::floor(argnum(a, 0, "floor")))); });
    register_builtin("math.ceil", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(std::ceil(argnum(a, 0, "ceil")))); });
    register_builtin("math.deg", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(argnum(a, 0, "deg") / (3.14159265358979323846 / 180.0))); });
    register_builtin("math.rad", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(argnum(a, 0, "rad") * (3.14159265358979323846 / 180.0))); });
    register_builtin("math.modf", [](Interp &, const Values &a, Values &r) {
        double x = argnum(a, 0, "modf"), ip = std::trunc(x);
        r.push_back(Value::number(ip));
        r.push_back(Value::number(std::isinf(x) ? std::copysign(0.0, x) : x - ip));
    });
    register_builtin("math.max", [](Interp &, const Values &a, Values &r) {
        double m = argnum(a, 0, "max");
        for (size_t i = 1; i < a.size(); ++i) { double d = argnum(a, i, "max"); if (d > m) m = d; }
        r.push_back(Value::number(m));
    });
    register_builtin("math.min", [](Interp &, const Values &a, Values &r) {
        double m = argnum(a, 0, "min");
        for (size_t i = 1; i < a.size(); ++i) { double d = argnum(a, i, "min"); if (d < m) m = d; }
        r.push_back(Value::number(m));
    });
    get_global("math").tab()->shash["pi"] = Value::number(3.14159265358979323846);
    get_global("math").tab()->shash["huge"] = Value::number(HUGE_VAL);

    register_builtin("table.unpack", [](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'unpack' (table expected)");
        double i = a.size() > 1 && a[1].t != Value::NIL ? argnum(a, 1, "unpack") : 1;
        double e = a.size() > 2 && a[2].t != Value::NIL ? argnum(a, 2, "unpack") : (double)a[0].tab()->length();
        for (double k = i; k <= e; k += 1) r.push_back(a[0].tab()->get(Value::number(k)));
    });
    register_builtin("table.insert", [](Interp &, const Values &a, Values &) {
        if (a.size() < 2 || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'insert' (table expected)");
        Table &t = *a[0].tab();
        if (a.size() == 2) { t.set(Value::number((double)t.length() + 1), a[1]); return; }
        size_t pos = (size_t)argnum(a, 1, "insert");
        if (pos < 1 || pos > t.length() + 1) throw LuaError("bad argument #2 to 'insert' (position out of bounds)");
        t.arr.insert(t.arr.begin() + (pos - 1), a[2]);
    });
    register_builtin("print", [](Interp &I, const Values &a, Values &) {
        std::string line;
        for (size_t i = 0; i < a.size(); ++i) { if (i) line += "\t"; line += I.tostring(a[i]); }
        if (I.print_sink) I.print_sink(line + "\n");
    });
    register_builtin("tostring", [](Interp &I, const Values &a, Values &r) { r.push_back(Value::string(I.tostring(a.empty() ? Value() : a[0]))); });
    register_builtin("tonumber", [](Interp &, const Values &a, Values &r) {
        double d;
        if (!a.empty() && Exec::tonumber(a[0], &d)) r.push_back(Value::number(d)); else r.push_back(Value());
    });
    register_builtin("type", [](Interp &, const Values &a, Values &r) {
        if (a.empty()) throw LuaError("bad argument #1 to 'type' (value expected)");
        r.push_back(Value::string(a[0].type_name()));
    });
    register_builtin("assert", [](Interp &I, const Values &a, Values &r) {
        if (a.empty() || !a[0].truthy()) throw LuaError(a.size() > 1 ? I.tostring(a[1]) : "assertion failed!");
        r = a;
    });
    // luaB_error: a string message gets the position of the call in front (level 1, the default; level 0 = none)
    register_builtin("error", [](Interp &I, const Values &a, Values &) {
        if (a.empty()) throw LuaError("nil");
        const bool positioned = a[0].t == Value::STR && !(a.size() > 1 && a[1].t == Value::NUM && a[1].n == 0) && I.call_chunk;
        throw LuaError((positioned ? *I.call_chunk + ":" + std::to_string(I.call_line) + ": " : std::string()) + I.tostring(a[0]));
    });
    register_builtin("select", [](Interp &, const Values &a, Values &r) {
        if (!a.empty() && a[0].t == Value::STR && a[0].str() == "#") { r.push_back(Value::number((double)a.size() - 1)); return; }
        double n = argnum(a, 0, "select");
        const double count = (double)a.size() - 1;
        if (n < 0) n = count + n + 1;                       // from the end (luaB_select)
        else if (n > count) n = count + 1;
        if (n < 1 || n != n) throw LuaError("bad argument #1 to 'select' (index out of range)");
        for (size_t i = (size_t)n; i < a.size(); ++i) r.push_back(a[i]);
    });
    // ---- chunks from strings and files (lbaselib.c load / loadfile / dofile, loadlib.c require): the reference opens the whole
    // library, so a lens script may keep shared helpers in a file of its own; paths are relative to the working directory, as there
    auto make_chunk = [](const std::string &src, const std::string &name) {
        std::shared_ptr<Chunk> ch = parse(src, name);
        auto cl = std::make_shared<Closure>();
        cl->proto = ch->main();
        cl->chunk = ch;
        return Value::closure(std::move(cl));
    };
    auto read_file = [](const std::string &path, std::string *out) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        char buf[4096];
        size_t n;
        out->clear();
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
        fclose(f);
        if (out->size() >= 1 && (*out)[0] == '#') { const size_t eol = out->find('\n'); out->replace(0, eol == std::string::npos ? out->size() : eol, ""); }   // (a first line starting with # is skipped, lauxlib.c)
        return true;
    };
    auto short_name = [](const std::string &path) { const size_t s = path.find_last_of('/'); return s == std::string::npos ? path : path.substr(s + 1); };
    register_builtin("load", [make_chunk](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'load' (string expected; reader functions are not supported)");
        const std::string name = a.size() > 1 && a[1].t == Value::STR ? a[1].str() : "=(load)";
        try { r.push_back(make_chunk(a[0].str(), name[0] == '=' || name[0] == '@' ? name.substr(1) : name)); }
        catch (const LuaError &e) { r.clear(); r.push_back(Value()); r.push_back(Value::string(e.what())); }
    });
    register_builtin("loadfile", [make_chunk, read_file, short_name](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'loadfile' (string expected)");
        std::string src;
        if (!read_file(a[0].str(), &src)) { r.push_back(Value()); r.push_back(Value::string("cannot open " + a[0].str())); return; }
        try { r.push_back(make_chunk(src, short_name(a[0].str()))); }
        catch (const LuaError &e) { r.clear(); r.push_back(Value()); r.push_back(Value::string(e.what())); }
    });
    register_builtin("dofile", [make_chunk, read_file, short_name](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'dofile' (string expected)");
        std::string src;
        if (!read_file(a[0].str(), &src)) throw LuaError("cannot open " + a[0].str());
        r = I.call(make_chunk(src, short_name(a[0].str())), Values());
    });
    {
        Value pkg = Value::table(std::make_shared<Table>());
        pkg.tab()->set(Value::string("loaded"), Value::table(std::make_shared<Table>()));
        const char *env = getenv("LUA_PATH_5_2");
        if (!env) env = getenv("LUA_PATH");
        pkg.tab()->set(Value::string("path"), Value::string(env ? env : "./?.lua;./?/init.lua"));
        globals["package"] = pkg;
    }
    register_builtin("require", [make_chunk, read_file, short_name](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'require' (string expected)");
        const std::string name = a[0].str();
        Value pkg = I.get_global("package");
        Value loaded = pkg.t == Value::TABLE ? pkg.tab()->get(Value::string("loaded")) : Value();
        if (loaded.t != Value::TABLE) throw LuaError("'package.loaded' must be a table");
        Value have = loaded.tab()->get(Value::string(name));
        if (have.truthy()) { r.push_back(have); return; }
        {                                                   // package.preload[name]: a loader the script registered itself (loadlib.c: searcher_preload)
            const Value preload = pkg.tab()->get(Value::string("preload"));
            const Value loader = preload.t == Value::TABLE ? preload.tab()->get(Value::string(name)) : Value();
            if (loader.t != Value::NIL) {
                Values out = I.call(loader, Values{Value::string(name)});
                Value result = !out.empty() && out[0].t != Value::NIL ? out[0] : Value();
                Value now = loaded.tab()->get(Value::string(name));
                if (result.t == Value::NIL) result = now.t != Value::NIL ? now : Value::boolean(true);
                loaded.tab()->set(Value::string(name), result);
                r.push_back(result);
                return;
            }
        }
        const Value pathv = pkg.tab()->get(Value::string("path"));
        if (pathv.t != Value::STR) throw LuaError("'package.path' must be a string");
        std::string file = name, tried;
        for (char &c : file) if (c == '.') c = '/';
        const std::string path = pathv.str();
        for (size_t at = 0; at <= path.size();) {
            size_t semi = path.find(';', at);
            if (semi == std::string::npos) semi = path.size();
            std::string candidate = path.substr(at, semi - at);
            at = semi + 1;
            if (candidate.empty()) continue;
            for (size_t q; (q = candidate.find('?')) != std::string::npos;) candidate.replace(q, 1, file);
            std::string src;
            if (!read_file(candidate, &src)) { tried += "\n\tno file '" + candidate + "'"; continue; }
            Values out = I.call(make_chunk(src, short_name(candidate)), Values{Value::string(name), Value::string(candidate)});
            Value result = !out.empty() && out[0].t != Value::NIL ? out[0] : Value();
            Value now = loaded.tab()->get(Value::string(name));                 // (the module may have set package.loaded[name] itself)
            if (result.t == Value::NIL) result = now.t != Value::NIL ? now : Value::boolean(true);
            loaded.tab()->set(Value::string(name), result);
            r.push_back(result);
            return;
        }
        throw LuaError("module '" + name + "' not found:" + tried);
    });
    // ---- bit32 (lbitlib.c): numbers taken modulo 2^32, results in [0, 2^32) -------------------------------------------------------
    auto bits = [](const Values &a, size_t i, const char *fn) -> uint32_t {
        const double d = argnum(a, i, fn);
        const double m = std::fmod(std::floor(d), 4294967296.0);                 // lua_Unsigned conversion: wraps around
        return (uint32_t)(m < 0 ? m + 4294967296.0 : m);
    };
    auto fold = [bits](const char *name, uint32_t start, uint32_t (*op)(uint32_t, uint32_t)) {
        return [bits, name, start, op](Interp &, const Values &a, Values &r) {
            uint32_t v = start;
            for (size_t i = 0; i < a.size(); ++i) v = op(v, bits(a, i, name));
            r.push_back(Value::number((double)v));
        };
    };
    register_builtin("bit32.band", fold("band", 0xFFFFFFFFu, [](uint32_t x, uint32_t y) { return x & y; }));
    register_builtin("bit32.bor", fold("bor", 0u, [](uint32_t x, uint32_t y) { return x | y; }));
    register_builtin("bit32.bxor", fold("bxor", 0u, [](uint32_t x, uint32_t y) { return x ^ y; }));
    register_builtin("bit32.btest", [bits](Interp &, const Values &a, Values &r) {
        uint32_t v = 0xFFFFFFFFu;
        for (size_t i = 0; i < a.size(); ++i) v &= bits(a, i, "btest");
        r.push_back(Value::boolean(v != 0));
    });
    register_builtin("bit32.bnot", [bits](Interp &, const Values &a, Values &r) { r.push_back(Value::number((double)(uint32_t)~bits(a, 0, "bnot"))); });
    auto shift = [](uint32_t v, long n) -> uint32_t {                             // n > 0: left; |n| >= 32 gives 0
        if (n <= -32 || n >= 32) return 0;
        return n >= 0 ? v << n : v >> -n;
    };
    register_builtin("bit32.lshift", [bits, shift](Interp &, const Values &a, Values &r) { r.push_back(Value::number((double)shift(bits(a, 0, "lshift"), (long)argnum(a, 1, "lshift")))); });
    register_builtin("bit32.rshift", [bits, shift](Interp &, const Values &a, Values &r) { r.push_back(Value::number((double)shift(bits(a, 0, "rshift"), -(long)argnum(a, 1, "rshift")))); });
    register_builtin("bit32.arshift", [bits, shift](Interp &, const Values &a, Values &r) {
        const uint32_t v = bits(a, 0, "arshift");
        const long n = (long)argnum(a, 1, "arshift");
        if (n < 0 || !(v & 0x80000000u)) { r.push_back(Value::number((double)shift(v, -n))); return; }
        r.push_back(Value::number((double)(n >= 32 ? 0xFFFFFFFFu : ((v >> n) | ~(0xFFFFFFFFu >> n)))));
    });
    register_builtin("bit32.lrotate", [bits](Interp &, const Values &a, Values &r) {
        const uint32_t v = bits(a, 0, "lrotate");
        const unsigned n = (unsigned)(((long)argnum(a, 1, "lrotate") % 32 + 32) % 32);
        r.push_back(Value::number((double)(n ? (v << n) | (v >> (32 - n)) : v)));
    });
    register_builtin("bit32.rrotate", [bits](Interp &, const Values &a, Values &r) {
        const uint32_t v = bits(a, 0, "rrotate");
        const unsigned n = (unsigned)(((-(long)argnum(a, 1, "rrotate")) % 32 + 32) % 32);
        r.push_back(Value::number((double)(n ? (v << n) | (v >> (32 - n)) : v)));
    });
    register_builtin("bit32.extract", [bits](Interp &, const Values &a, Values &r) {
        const uint32_t v = bits(a, 0, "extract");
        const long f = (long)argnum(a, 1, "extract"), w = a.size() > 2 && a[2].t != Value::NIL ? (long)argnum(a, 2, "extract") : 1;
        if (f < 0) throw LuaError("bad argument #2 to 'extract' (field cannot be negative)");
        if (w <= 0) throw LuaError("bad argument #3 to 'extract' (width must be positive)");
        if (f + w > 32) throw LuaError("trying to access non-existent bits");
        r.push_back(Value::number((double)((v >> f) & (w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1)))));
    });
    register_builtin("bit32.replace", [bits](Interp &, const Values &a, Values &r) {
        const uint32_t v = bits(a, 0, "replace"), u = bits(a, 1, "replace");
        const long f = (long)argnum(a, 2, "replace"), w = a.size() > 3 && a[3].t != Value::NIL ? (long)argnum(a, 3, "replace") : 1;
        if (f < 0) throw LuaError("bad argument #3 to 'replace' (field cannot be negative)");
        if (w <= 0) throw LuaError("bad argument #4 to 'replace' (width must be positive)");
        if (f + w > 32) throw LuaError("trying to access non-existent bits");
        const uint32_t m = (w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1)) << f;
        r.push_back(Value::number((double)((v & ~m) | ((u << f) & m))));
    });
    register_builtin("os.time", [](Interp &, const Values &, Values &r) { r.push_back(Value::number((double)time(nullptr))); });
    register_builtin("os.clock", [](Interp &, const Values &, Values &r) { r.push_back(Value::number((double)clock() / (double)CLOCKS_PER_SEC)); });
    register_builtin("os.getenv", [](Interp &, const Values &a, Values &r) {
        const char *v = !a.empty() && a[0].t == Value::STR ? getenv(a[0].str().c_str()) : nullptr;
        r.push_back(v ? Value::string(v) : Value());
    });
    // io.open / io.lines and the file methods read / lines / write / close (liolib.c): enough for a script that reads a table of
    // numbers - a measured lens profile - while it loads.  A file object is a table of its methods over one shared FILE.
    struct FileBox { FILE *f = nullptr; ~FileBox() { if (f) fclose(f); } };
    auto read_one = [](FileBox &fb, const Value &fmt, Values &r) -> bool {       // one format of file:read; false at end of file
        if (!fb.f) throw LuaError("attempt to use a closed file");
        if (fmt.t == Value::NUM) {
            // (liolib.c reads up to n bytes; n comes from the script: not negative, not NaN, and no more than the file can hold)
            if (!(fmt.n >= 0) || fmt.n > 1073741824.0) throw LuaError("bad argument #1 to 'read' (invalid count)");
            std::string out((size_t)fmt.n, '\0');
            const size_t got = fread(&out[0], 1, out.size(), fb.f);
            if (got == 0 && out.size() > 0) { r.push_back(Value()); return false; }
            out.resize(got);
            r.push_back(Value::string(out));
            return true;
        }
        std::string what = fmt.t == Value::STR ? fmt.str() : "";
        if (!what.empty() && what[0] == '*') what.erase(0, 1);
        if (what == "n") {
            double v;
            if (fscanf(fb.f, "%lf", &v) != 1) { r.push_back(Value()); return false; }
            r.push_back(Value::number(v));
            return true;
        }
        if (what == "a") {
            std::string out;
            char buf[4096];
            size_t n;
            while ((n = fread(buf, 1, sizeof buf, fb.f)) > 0) out.append(buf, n);
            r.push_back(Value::string(out));
            return true;
        }
        if (what == "l" || what == "L") {
            std::string out;
            int c;
            bool any = false;
            while ((c = fgetc(fb.f)) != EOF) { any = true; if (c == '\n') { if (what == "L") out += '\n'; break; } out += (char)c; }
            if (!any) { r.push_back(Value()); return false; }
            r.push_back(Value::string(out));
            return true;
        }
        throw LuaError("bad argument to 'read' (invalid format)");
    };
    auto make_file = [read_one](FILE *fp) {
        auto fb = std::make_shared<FileBox>();
        fb->f = fp;
        Value obj = Value::table(std::make_shared<Table>());
        auto method = [&obj](const char *name, BuiltinFn fn) {
            Value v;
            v.t = Value::BUILTIN;
            auto bp = std::make_shared<Builtin>();
            bp->name = std::string("file:") + name;
            bp->fn = std::move(fn);
            v.p = std::move(bp);
            obj.tab()->set(Value::string(name), v);
        };
        method("read", [fb, read_one](Interp &, const Values &a, Values &r) {
            if (a.size() <= 1) { read_one(*fb, Value::string("l"), r); return; }
            for (size_t i = 1; i < a.size(); ++i) if (!read_one(*fb, a[i], r)) break;
        });
        method("lines", [fb, read_one](Interp &, const Values &, Values &r) {
            Value it;
            it.t = Value::BUILTIN;
            auto bp = std::make_shared<Builtin>();
            bp->name = "file lines iterator";
            bp->fn = [fb, read_one](Interp &, const Values &, Values &out) { read_one(*fb, Value::string("l"), out); };
            it.p = std::move(bp);
            r.push_back(it);
        });
        method("write", [fb](Interp &I, const Values &a, Values &r) {
            if (!fb->f) throw LuaError("attempt to use a closed file");
            for (size_t i = 1; i < a.size(); ++i) {
                if (a[i].t != Value::STR && a[i].t != Value::NUM) throw LuaError(std::string("bad argument to 'write' (string expected, got ") + a[i].type_name() + ")");
                const std::string t = I.tostring(a[i]);
                fwrite(t.data(), 1, t.size(), fb->f);
            }
            r.push_back(a.empty() ? Value() : a[0]);
        });
        method("close", [fb](Interp &, const Values &, Values &r) {
            if (fb->f) { fclose(fb->f); fb->f = nullptr; }
            r.push_back(Value::boolean(true));
        });
        method("flush", [fb](Interp &, const Values &a, Values &r) {
            if (!fb->f) throw LuaError("attempt to use a closed file");
            fflush(fb->f);
            r.push_back(a.empty() ? Value() : a[0]);
        });
        method("seek", [fb](Interp &, const Values &a, Values &r) {              // f_seek (liolib.c): whence "set" / "cur" / "end", offset
            if (!fb->f) throw LuaError("attempt to use a closed file");
            const std::string whence = a.size() > 1 && a[1].t == Value::STR ? a[1].str() : "cur";
            const int w = whence == "set" ? SEEK_SET : whence == "cur" ? SEEK_CUR : whence == "end" ? SEEK_END : -1;
            if (w < 0) throw LuaError("bad argument #1 to 'seek' (invalid option '" + whence + "')");
            const double off = a.size() > 2 && a[2].t != Value::NIL ? argnum(a, 2, "seek") : 0.0;
            if (fseek(fb->f, (long)off, w) != 0) { const int e = errno; r.push_back(Value()); r.push_back(Value::string(strerror(e))); r.push_back(Value::number((double)e)); return; }
            r.push_back(Value::number((double)ftell(fb->f)));
        });
        method("setvbuf", [](Interp &, const Values &, Values &r) { r.push_back(Value::boolean(true)); });
        return obj;
    };
    register_builtin("io.open", [make_file](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'open' (string expected)");
        const std::string mode = a.size() > 1 && a[1].t == Value::STR ? a[1].str() : "r";
        if (mode.empty() || !strchr("rwa", mode[0]) || mode.find_first_not_of("rwa+b") != std::string::npos) throw LuaError("bad argument #2 to 'open' (invalid mode)");
        FILE *fp = fopen(a[0].str().c_str(), mode.c_str());
        if (!fp) { r.push_back(Value()); r.push_back(Value::string(a[0].str() + ": " + strerror(errno))); r.push_back(Value::number((double)errno)); return; }
        r.push_back(make_file(fp));
    });
    register_builtin("io.lines", [make_file](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'lines' (file name expected)");
        FILE *fp = fopen(a[0].str().c_str(), "r");
        if (!fp) throw LuaError(a[0].str() + ": " + strerror(errno));
        Value file = make_file(fp);
        Values it = I.call(file.tab()->get(Value::string("lines")), Values{file});
        r.push_back(it[0]);                                                      // (the file closes when the iterator is collected)
    });
    register_builtin("io.write", [](Interp &I, const Values &a, Values &) {       // to where print goes, without its tabs and newline
        std::string out;
        for (const Value &v : a) {
            if (v.t != Value::STR && v.t != Value::NUM) throw LuaError(std::string("bad argument to 'write' (string expected, got ") + v.type_name() + ")");
            out += I.tostring(v);
        }
        if (I.print_sink) I.print_sink(out);
    });
    // (r6) the rest of what luaL_openlibs (fisheye.c:1224) gives a script while it LOADS, short of coroutines: os.date / difftime / remove /
    // rename / tmpname / setlocale / exit, io.read / close / flush / input / output / stdin / stdout / stderr, table.pack, xpcall,
    // collectgarbage, _VERSION, _G, package.preload, debug.traceback / getinfo, string.dump (refuses, as for a C function)
    register_builtin("os.difftime", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(argnum(a, 0, "difftime") - (a.size() > 1 && a[1].t != Value::NIL ? argnum(a, 1, "difftime") : 0.0))); });
    register_builtin("os.date", [](Interp &, const Values &a, Values &r) {
        std::string fmt = !a.empty() && a[0].t == Value::STR ? a[0].str() : "%c";
        const time_t t = a.size() > 1 && a[1].t != Value::NIL ? (time_t)argnum(a, 1, "date") : time(nullptr);
        bool utc = false;
        if (!fmt.empty() && fmt[0] == '!') { utc = true; fmt.erase(0, 1); }
        struct tm tmv;
        if (!(utc ? gmtime_r(&t, &tmv) : localtime_r(&t, &tmv))) { r.push_back(Value()); return; }
        if (fmt.compare(0, 2, "*t") == 0) {
            Value tv = Value::table(std::make_shared<Table>());
            auto setn = [&](const char *k, double v) { tv.tab()->set(Value::string(k), Value::number(v)); };
            setn("sec", tmv.tm_sec); setn("min", tmv.tm_min); setn("hour", tmv.tm_hour); setn("day", tmv.tm_mday); setn("month", tmv.tm_mon + 1);
            setn("year", tmv.tm_year + 1900); setn("wday", tmv.tm_wday + 1); setn("yday", tmv.tm_yday + 1);
            tv.tab()->set(Value::string("isdst"), Value::boolean(tmv.tm_isdst > 0));
            r.push_back(tv);
            return;
        }
        std::string out;
        for (size_t i = 0; i < fmt.size(); ++i) {
            if (fmt[i] != '%') { out += fmt[i]; continue; }
            if (i + 1 >= fmt.size() || !strchr("aAbBcdHIjmMpSUwWxXyYZ%", fmt[i + 1])) throw LuaError("bad argument #1 to 'date' (invalid conversion specifier '%" + fmt.substr(i + 1, 1) + "')");
            const char spec[3] = {'%', fmt[++i], 0};
            char buf[200];
            out.append(buf, strftime(buf, sizeof buf, spec, &tmv));
        }
        r.push_back(Value::string(out));
    });
    auto os_result = [](bool ok, const std::string &name, Values &r) {              // os_pushresult (loslib.c)
        if (ok) { r.push_back(Value::boolean(true)); return; }
        const int e = errno;
        r.push_back(Value()); r.push_back(Value::string(name + ": " + strerror(e))); r.push_back(Value::number((double)e));
    };
    register_builtin("os.remove", [os_result](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::STR) throw LuaError("bad argument #1 to 'remove' (string expected)");
        os_result(remove(a[0].str().c_str()) == 0, a[0].str(), r);
    });
    register_builtin("os.rename", [os_result](Interp &, const Values &a, Values &r) {
        if (a.size() < 2 || a[0].t != Value::STR || a[1].t != Value::STR) throw LuaError("bad argument to 'rename' (string expected)");
        os_result(rename(a[0].str().c_str(), a[1].str().c_str()) == 0, a[0].str(), r);
    });
    register_builtin("os.tmpname", [](Interp &, const Values &, Values &r) {
        char name[] = "/tmp/lua_XXXXXX";
        const int fd = mkstemp(name);
        if (fd < 0) throw LuaError("unable to generate a unique filename");
        close(fd);
        r.push_back(Value::string(name));
    });
    register_builtin("os.setlocale", [](Interp &, const Values &a, Values &r) {     // the library runs in the "C" locale and leaves the process's alone
        const bool c = a.empty() || a[0].t == Value::NIL || (a[0].t == Value::STR && (a[0].str().empty() || a[0].str() == "C" || a[0].str() == "POSIX"));
        r.push_back(c ? Value::string("C") : Value());
    });
    // (the reference's VM would end the engine's process; a library does not: the script gets an error it can see)
    register_builtin("os.exit", [](Interp &, const Values &, Values &) { throw LuaError("os.exit: a lens / globe script cannot end the host process"); });
    register_builtin("table.pack", [](Interp &, const Values &a, Values &r) {
        Value t = Value::table(std::make_shared<Table>());
        for (size_t i = 0; i < a.size(); ++i) t.tab()->set(Value::number((double)i + 1), a[i]);
        t.tab()->set(Value::string("n"), Value::number((double)a.size()));
        r.push_back(t);
    });
    register_builtin("collectgarbage", [](Interp &, const Values &a, Values &r) {   // reference counting: nothing to run; "count" answers 0 KB
        const std::string opt = !a.empty() && a[0].t == Value::STR ? a[0].str() : "collect";
        static const char *const known[] = {"collect", "stop", "restart", "count", "step", "setpause", "setstepmul", "isrunning", "generational", "incremental"};
        bool ok = false;
        for (const char *k : known) ok = ok || opt == k;
        if (!ok) throw LuaError("bad argument #1 to 'collectgarbage' (invalid option '" + opt + "')");
        if (opt == "isrunning" || opt == "step") r.push_back(Value::boolean(true));
        else { r.push_back(Value::number(0)); if (opt == "count") r.push_back(Value::number(0)); }
    });
    // xpcall(f, msgh, ...): pcall with the message handed to msgh first (no traceback to unwind here: msgh sees the message)
    register_builtin("xpcall", [](Interp &I, const Values &a, Values &r) {
        if (a.size() < 2) throw LuaError("bad argument #2 to 'xpcall' (value expected)");
        Values args;
        args.append(a.begin() + 2, a.end());
        const int depth = I.depth;
        try {
            Values out = I.call(a[0], args);
            r.push_back(Value::boolean(true));
            r.append(out.begin(), out.end());
        } catch (const LuaError &e) {
            I.depth = depth;
            r.clear();
            r.push_back(Value::boolean(false));
            Values h = I.call(a[1], Values{Value::string(e.what())});
            r.append(h.begin(), h.end());
        }
    });
    register_builtin("debug.traceback", [](Interp &I, const Values &a, Values &r) {
        if (!a.empty() && a[0].t != Value::STR && a[0].t != Value::NIL && a[0].t != Value::NUM) { r.push_back(a[0]); return; }   // (a non-string message is returned as it is)
        r.push_back(Value::string((a.empty() || a[0].t == Value::NIL ? std::string() : I.tostring(a[0]) + "\n") + "stack traceback:\n\t[host interpreter: no frames recorded]"));
    });
    register_builtin("debug.getinfo", [](Interp &I, const Values &, Values &r) {
        Value t = Value::table(std::make_shared<Table>());
        t.tab()->set(Value::string("currentline"), Value::number((double)I.call_line));
        t.tab()->set(Value::string("short_src"), Value::string(I.call_chunk ? *I.call_chunk : std::string("?")));
        t.tab()->set(Value::string("source"), Value::string("@" + (I.call_chunk ? *I.call_chunk : std::string("?"))));
        t.tab()->set(Value::string("what"), Value::string("Lua"));
        r.push_back(t);
    });
    // the rest of ldblib.c, as far as a tree-walking interpreter without a register stack can honour it: metatables without the
    // __metatable guard, a closure's upvalues by number (the cells themselves: setupvalue / upvaluejoin act on what the closure sees),
    // hooks that are accepted and never fire, locals that are not there to be had (nil), the registry as one table per state
    register_builtin("debug.getmetatable", [](Interp &, const Values &a, Values &r) {
        r.push_back(!a.empty() && a[0].t == Value::TABLE && a[0].tab()->meta ? Value::table(a[0].tab()->meta) : Value());
    });
    register_builtin("debug.setmetatable", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 2 || (a[1].t != Value::NIL && a[1].t != Value::TABLE)) throw LuaError("bad argument #2 to 'setmetatable' (nil or table expected)");
        if (a[0].t == Value::TABLE) a[0].tab()->meta = a[1].t == Value::TABLE ? a[1].tab_ptr() : nullptr;
        else if (a[1].t != Value::NIL) throw LuaError("debug.setmetatable: only tables carry a metatable in this interpreter");
        r.push_back(a[0]);
    });
    register_builtin("debug.getupvalue", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 2 || (a[0].t != Value::FUNC && a[0].t != Value::BUILTIN)) throw LuaError("bad argument #1 to 'getupvalue' (function expected)");
        if (a[1].t != Value::NUM) throw LuaError("bad argument #2 to 'getupvalue' (number expected)");
        if (a[0].t != Value::FUNC) return;                                  // (a C function here has none)
        const Closure *c = a[0].fn();
        const double k = a[1].n;
        if (!(k >= 1 && k <= (double)c->upvals.size()) || k != (double)(size_t)k) return;
        const size_t i = (size_t)k - 1;
        r.push_back(Value::string(i < c->proto->upvals.size() ? c->proto->upvals[i].name : std::string("?")));
        r.push_back(c->upvals[i] ? *c->upvals[i] : Value());
    });
    register_builtin("debug.setupvalue", [](Interp &I, const Values &a, Values &r) {
        if (a.size() < 3 || a[0].t != Value::FUNC) throw LuaError("bad argument #1 to 'setupvalue' (Lua function expected)");
        if (a[1].t != Value::NUM) throw LuaError("bad argument #2 to 'setupvalue' (number expected)");
        Closure *c = a[0].fn();
        const double k = a[1].n;
        if (!(k >= 1 && k <= (double)c->upvals.size()) || k != (double)(size_t)k) return;
        const size_t i = (size_t)k - 1;
        if (!c->upvals[i]) c->upvals[i] = std::make_shared<Value>();
        *c->upvals[i] = a[2];
        ++I.activity;
        r.push_back(Value::string(i < c->proto->upvals.size() ? c->proto->upvals[i].name : std::string("?")));
    });
    register_builtin("debug.upvalueid", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 2 || a[0].t != Value::FUNC || a[1].t != Value::NUM) throw LuaError("bad argument to 'upvalueid' (Lua function, number expected)");
        const Closure *c = a[0].fn();
        const double k = a[1].n;
        if (!(k >= 1 && k <= (double)c->upvals.size())) throw LuaError("bad argument #2 to 'upvalueid' (invalid upvalue index)");
        r.push_back(Value::number((double)(reinterpret_cast<uintptr_t>(c->upvals[(size_t)k - 1].get()) >> 3)));   // (a light userdata there: comparable, nothing more)
    });
    register_builtin("debug.upvaluejoin", [](Interp &I, const Values &a, Values &) {
        if (a.size() < 4 || a[0].t != Value::FUNC || a[2].t != Value::FUNC || a[1].t != Value::NUM || a[3].t != Value::NUM)
            throw LuaError("bad argument to 'upvaluejoin' (Lua function, number, Lua function, number expected)");
        Closure *c1 = a[0].fn(), *c2 = a[2].fn();
        const double k1 = a[1].n, k2 = a[3].n;
        if (!(k1 >= 1 && k1 <= (double)c1->upvals.size()) || !(k2 >= 1 && k2 <= (double)c2->upvals.size())) throw LuaError("bad argument to 'upvaluejoin' (invalid upvalue index)");
        c1->upvals[(size_t)k1 - 1] = c2->upvals[(size_t)k2 - 1];
        ++I.activity;
    });
    register_builtin("debug.getlocal", [](Interp &, const Values &, Values &r) { r.push_back(Value()); });          // (no frame keeps its locals by name)
    register_builtin("debug.setlocal", [](Interp &, const Values &, Values &r) { r.push_back(Value()); });
    register_builtin("debug.gethook", [](Interp &, const Values &, Values &r) { r.push_back(Value()); });           // (none is ever set)
    register_builtin("debug.sethook", [](Interp &, const Values &, Values &) {});                                   // (accepted, never called)
    register_builtin("debug.getuservalue", [](Interp &, const Values &, Values &r) { r.push_back(Value()); });      // (no userdata in this interpreter)
    register_builtin("debug.setuservalue", [](Interp &, const Values &a, Values &r) { r.push_back(a.empty() ? Value() : a[0]); });
    register_builtin("debug.debug", [](Interp &, const Values &, Values &) {});                                     // (no console to read commands from)
    register_builtin("debug.getregistry", [](Interp &I, const Values &, Values &r) {                               // (one table per interpreter state)
        if (I.registry.t != Value::TABLE) I.registry = Value::table(std::make_shared<Table>());
        r.push_back(I.registry);
    });
    register_builtin("string.dump", [](Interp &, const Values &, Values &) { throw LuaError("unable to dump given function"); });
    globals["_VERSION"] = Value::string("Lua 5.2");
    {
        // _G: the globals live in the interpreter's own map, not in a Lua table - _G is a proxy whose __index / __newindex reach them
        // (_G.x, _G[name] = v, rawget / pairs on it see nothing: the one difference from lbaselib.c's table of globals)
        Value g = Value::table(std::make_shared<Table>()), mt = Value::table(std::make_shared<Table>());
        auto fn = [](const char *name, BuiltinFn f) { Value v; v.t = Value::BUILTIN; auto bp = std::make_shared<Builtin>(); bp->name = name; bp->fn = std::move(f); v.p = std::move(bp); return v; };
        mt.tab()->set(Value::string("__index"), fn("_G.__index", [](Interp &I, const Values &a, Values &r) {
            r.push_back(a.size() > 1 && a[1].t == Value::STR ? I.get_global(a[1].str()) : Value());
        }));
        mt.tab()->set(Value::string("__newindex"), fn("_G.__newindex", [](Interp &I, const Values &a, Values &) {
            if (a.size() < 3 || a[1].t != Value::STR) throw LuaError("_G: global names are strings");
            I.set_global(a[1].str(), a[2]);
        }));
        g.tab()->meta = mt.tab_ptr();
        globals["_G"] = g;
    }
    if (get_global("package").t == Value::TABLE && get_global("package").tab()->get(Value::string("preload")).t == Value::NIL)
        get_global("package").tab()->set(Value::string("preload"), Value::table(std::make_shared<Table>()));
    // io.stdin / io.stdout / io.stderr and the default files of io.read / io.write (liolib.c): stdout is where print goes (the host's console)
    {
        auto std_file = [make_file, read_one](FILE *fp, bool to_console) {
            Value f = make_file(nullptr);                    // (methods over a box that owns nothing: the standard streams are never closed)
            auto named = [&f](const char *name, BuiltinFn fn) {
                Value v; v.t = Value::BUILTIN;
                auto bp = std::make_shared<Builtin>(); bp->name = std::string("file:") + name; bp->fn = std::move(fn); v.p = std::move(bp);
                f.tab()->set(Value::string(name), v);
            };
            named("write", [fp, to_console](Interp &I, const Values &a, Values &r) {
                std::string out;
                for (size_t i = 1; i < a.size(); ++i) {
                    if (a[i].t != Value::STR && a[i].t != Value::NUM) throw LuaError(std::string("bad argument to 'write' (string expected, got ") + a[i].type_name() + ")");
                    out += I.tostring(a[i]);
                }
                if (to_console) { if (I.print_sink) I.print_sink(out); } else fwrite(out.data(), 1, out.size(), fp);
                r.push_back(a.empty() ? Value() : a[0]);
            });
            named("read", [fp, read_one](Interp &, const Values &a, Values &r) {
                struct Borrow { FileBox fb; ~Borrow() { fb.f = nullptr; } } b;          // (read_one works on a FileBox; this one must not close the stream)
                b.fb.f = fp;
                if (a.size() <= 1) { read_one(b.fb, Value::string("l"), r); return; }
                for (size_t i = 1; i < a.size(); ++i) if (!read_one(b.fb, a[i], r)) break;
            });
            named("close", [](Interp &, const Values &, Values &r) { r.push_back(Value()); r.push_back(Value::string("cannot close standard file")); });
            named("flush", [fp, to_console](Interp &, const Values &a, Values &r) { if (!to_console) fflush(fp); r.push_back(a.empty() ? Value() : a[0]); });
            named("setvbuf", [](Interp &, const Values &, Values &r) { r.push_back(Value::boolean(true)); });
            return f;
        };
        Value io = get_global("io");
        const Value in = std_file(stdin, false), out = std_file(stdout, true), err = std_file(stderr, false);
        io.tab()->set(Value::string("stdin"), in);
        io.tab()->set(Value::string("stdout"), out);
        io.tab()->set(Value::string("stderr"), err);
        // the default input / output files: cells the io.* functions below share
        auto def_in = std::make_shared<Value>(in), def_out = std::make_shared<Value>(out);
        auto open_or_file = [make_file](const Values &a, const char *mode, const char *fn) -> Value {
            if (a[0].t == Value::STR) {
                FILE *fp = fopen(a[0].str().c_str(), mode);
                if (!fp) throw LuaError(a[0].str() + ": " + strerror(errno));
                return make_file(fp);
            }
            if (a[0].t == Value::TABLE && a[0].tab()->get(Value::string("read")).is_function()) return a[0];
            throw LuaError(std::string("bad argument #1 to '") + fn + "' (file expected)");
        };
        register_builtin("io.input", [def_in, open_or_file](Interp &, const Values &a, Values &r) {
            if (!a.empty() && a[0].t != Value::NIL) *def_in = open_or_file(a, "r", "input");
            r.push_back(*def_in);
        });
        register_builtin("io.output", [def_out, open_or_file](Interp &, const Values &a, Values &r) {
            if (!a.empty() && a[0].t != Value::NIL) *def_out = open_or_file(a, "w", "output");
            r.push_back(*def_out);
        });
        register_builtin("io.read", [def_in](Interp &I, const Values &a, Values &r) {
            Values args{*def_in};
            args.append(a.begin(), a.end());
            r = I.call(def_in->tab()->get(Value::string("read")), args);
        });
        register_builtin("io.write", [def_out](Interp &I, const Values &a, Values &r) {       // (the default output is the console until io.output says otherwise)
            Values args{*def_out};
            args.append(a.begin(), a.end());
            r = I.call(def_out->tab()->get(Value::string("write")), args);
        });
        register_builtin("io.close", [def_out](Interp &I, const Values &a, Values &r) {
            const Value f = !a.empty() && a[0].t == Value::TABLE ? a[0] : *def_out;
            r = I.call(f.tab()->get(Value::string("close")), Values{f});
        });
        register_builtin("io.flush", [def_out](Interp &I, const Values &, Values &r) {
            const Value fl = def_out->tab()->get(Value::string("flush"));
            if (fl.is_function()) r = I.call(fl, Values{*def_out});
        });
    }
    // ---- coroutines (lcorolib.c).  The interpreter walks the tree on the native stack, so a coroutine is a native stack of its own: a thread
    // that runs ONLY while its resumer waits (strict hand-over under the coroutine's mutex: the interpreter's state is never touched by two
    // threads at once).  create / resume / yield / status / running / wrap, yields across pcall and metamethods included (there is no C
    // boundary here to refuse them).  A coroutine that is dropped while suspended is unwound (its yield throws) before its thread is joined.
    struct CoKill {};
    struct Coroutine {
        Value fn;
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        enum { FRESH, RUNNING, SUSPENDED, NORMAL, DEAD } state = FRESH;
        bool co_turn = false, kill = false, failed = false;
        Values xfer;
        std::string err;
        int depth = 0;                                       // the interpreter's call depth inside this coroutine
        void to_co() { std::unique_lock<std::mutex> lock(m); co_turn = true; cv.notify_all(); cv.wait(lock, [this] { return !co_turn; }); }
        void to_resumer() { std::unique_lock<std::mutex> lock(m); co_turn = false; cv.notify_all(); cv.wait(lock, [this] { return co_turn; }); }
        ~Coroutine()
        {
            if (!th.joinable()) return;
            if (std::this_thread::get_id() == th.get_id()) { th.detach(); return; }      // (the last reference died inside its own body)
            if (state != DEAD) { kill = true; to_co(); }
            th.join();
        }
    };
    auto co_of = [](const Values &a, const char *fn) -> std::shared_ptr<Coroutine> {
        if (a.empty() || a[0].t != Value::THREAD) throw LuaError(std::string("bad argument #1 to '") + fn + "' (coroutine expected)");
        return std::static_pointer_cast<Coroutine>(a[0].p);
    };
    auto co_resume = [](Interp &I, const std::shared_ptr<Coroutine> &co, const Value *first, const Value *last, Values &r) -> bool {
        if (co->state == Coroutine::DEAD) { r.push_back(Value::string("cannot resume dead coroutine")); return false; }
        if (co->state == Coroutine::RUNNING || co->state == Coroutine::NORMAL) { r.push_back(Value::string("cannot resume non-suspended coroutine")); return false; }
        Coroutine *prev = static_cast<Coroutine *>(I.current_co);
        if (prev) prev->state = Coroutine::NORMAL;
        const int depth = I.depth;
        const bool fresh = co->state == Coroutine::FRESH;
        co->xfer.assign(first, last);
        co->state = Coroutine::RUNNING;
        I.current_co = co.get();
        I.depth = co->depth;
        if (fresh) {
            Coroutine *c = co.get();
            Interp *ip = &I;
            co->th = std::thread([c, ip]() {
                { std::unique_lock<std::mutex> lock(c->m); c->cv.wait(lock, [c] { return c->co_turn; }); }
                if (!c->kill) {
                    try { Values out = ip->call(c->fn, c->xfer); c->xfer = std::move(out); }
                    catch (const LuaError &e) { c->failed = true; c->err = e.what(); }
                    catch (const CoKill &) {}
                }
                c->state = Coroutine::DEAD;
                std::unique_lock<std::mutex> lock(c->m);
                c->co_turn = false;
                c->cv.notify_all();
            });
        }
        co->to_co();                                         // ... and wait until it yields, returns or fails
        co->depth = I.depth;
        I.depth = depth;
        I.current_co = prev;
        if (prev) prev->state = Coroutine::RUNNING;
        if (co->failed) { co->failed = false; r.push_back(Value::string(co->err)); return false; }
        r.append(co->xfer.begin(), co->xfer.end());
        return true;
    };
    register_builtin("coroutine.create", [](Interp &, const Values &a, Values &r) {
        if (a.empty() || !a[0].is_function()) throw LuaError("bad argument #1 to 'create' (function expected)");
        auto co = std::make_shared<Coroutine>();
        co->fn = a[0];
        Value v;
        v.t = Value::THREAD;
        v.p = co;
        r.push_back(v);
    });
    register_builtin("coroutine.resume", [co_of, co_resume](Interp &I, const Values &a, Values &r) {
        auto co = co_of(a, "resume");
        Values out;
        const bool ok = co_resume(I, co, a.begin() + 1, a.end(), out);
        r.push_back(Value::boolean(ok));
        r.append(out.begin(), out.end());
    });
    register_builtin("coroutine.yield", [](Interp &I, const Values &a, Values &r) {
        Coroutine *co = static_cast<Coroutine *>(I.current_co);
        if (!co) throw LuaError("attempt to yield from outside a coroutine");
        co->xfer = a;
        co->state = Coroutine::SUSPENDED;
        co->to_resumer();                                    // ... until somebody resumes (or drops) this coroutine
        if (co->kill) throw CoKill();
        r = co->xfer;                                        // what resume was given
    });
    register_builtin("coroutine.status", [co_of](Interp &, const Values &a, Values &r) {
        static const char *const names[] = {"suspended", "running", "suspended", "normal", "dead"};
        r.push_back(Value::string(names[co_of(a, "status")->state]));
    });
    register_builtin("coroutine.running", [](Interp &I, const Values &, Values &r) {
        // (the main thread has no object here: nil, true - Lua 5.2 returns the main coroutine and true)
        r.push_back(Value());
        r.push_back(Value::boolean(I.current_co == nullptr));
        (void)I;
    });
    register_builtin("coroutine.wrap", [co_resume](Interp &, const Values &a, Values &r) {
        if (a.empty() || !a[0].is_function()) throw LuaError("bad argument #1 to 'wrap' (function expected)");
        auto co = std::make_shared<Coroutine>();
        co->fn = a[0];
        Value v;
        v.t = Value::BUILTIN;
        auto bp = std::make_shared<Builtin>();
        bp->name = "coroutine.wrap";
        bp->fn = [co, co_resume](Interp &I, const Values &args, Values &out) {
            Values res;
            if (!co_resume(I, co, args.begin(), args.end(), res)) throw LuaError(!res.empty() && res[0].t == Value::STR ? res[0].str() : std::string("error in a wrapped coroutine"));
            out = std::move(res);
        };
        v.p = std::move(bp);
        r.push_back(v);
    });
    // pcall(f, ...): true + results, or false + the error message (lua_pcall; messages carry no traceback here either)
    register_builtin("pcall", [](Interp &I, const Values &a, Values &r) {
        if (a.empty()) throw LuaError("bad argument #1 to 'pcall' (value expected)");
        Values args;
        args.append(a.begin() + 1, a.end());
        const int depth = I.depth;
        try {
            Values out = I.call(a[0], args);
            r.push_back(Value::boolean(true));
            r.append(out.begin(), out.end());
        } catch (const LuaError &e) {
            I.depth = depth;
            r.clear();
            r.push_back(Value::boolean(false));
            r.push_back(Value::string(e.what()));
        }
    });
    register_builtin("setmetatable", [](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'setmetatable' (table expected)");
        if (a.size() < 2 || (a[1].t != Value::NIL && a[1].t != Value::TABLE)) throw LuaError("bad argument #2 to 'setmetatable' (nil or table expected)");
        Table &t = *a[0].tab();
        if (t.meta && t.meta->get(Value::string("__metatable")).t != Value::NIL) throw LuaError("cannot change a protected metatable");
        t.meta = a[1].t == Value::TABLE ? a[1].tab_ptr() : nullptr;
        r.push_back(a[0]);
    });
    register_builtin("getmetatable", [](Interp &I, const Values &a, Values &r) {
        if (!a.empty() && a[0].t == Value::TABLE && a[0].tab()->meta) {
            Value guard = a[0].tab()->meta->get(Value::string("__metatable"));
            r.push_back(guard.t != Value::NIL ? guard : Value::table(a[0].tab()->meta));
        } else if (!a.empty() && a[0].t == Value::STR) {               // every string shares one: { __index = string }
            Value mt = Value::table(std::make_shared<Table>());
            mt.tab()->set(Value::string("__index"), I.get_global("string"));
            r.push_back(mt);
        } else r.push_back(Value());
    });
    register_builtin("rawget", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 2 || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'rawget' (table expected)");
        r.push_back(a[0].tab()->get(a[1]));
    });
    register_builtin("rawset", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 3 || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'rawset' (table expected)");
        a[0].tab()->set(a[1], a[2]);
        r.push_back(a[0]);
    });
    register_builtin("rawequal", [](Interp &, const Values &a, Values &r) {
        if (a.size() < 2) throw LuaError("bad argument #2 to 'rawequal' (value expected)");
        r.push_back(Value::boolean(Exec::raw_equal(a[0], a[1])));
    });
    register_builtin("rawlen", [](Interp &, const Values &a, Values &r) {
        if (!a.empty() && a[0].t == Value::TABLE) r.push_back(Value::number((double)a[0].tab()->length()));
        else if (!a.empty() && a[0].t == Value::STR) r.push_back(Value::number((double)a[0].str().size()));
        else throw LuaError("table or string expected");
    });
    // ---- string library (lstrlib.c), without patterns: find takes plain strings only -------------------------------------------
    auto argstr = [](const Values &a, size_t i, const char *fn) -> std::string {
        if (i < a.size() && a[i].t == Value::STR) return a[i].str();
        if (i < a.size() && a[i].t == Value::NUM) { char b[64]; snprintf(b, sizeof b, "%.14g", a[i].n); return b; }
        throw LuaError(std::string("bad argument #") + std::to_string(i + 1) + " to '" + fn + "' (string expected, got " +
                       (i < a.size() ? a[i].type_name() : "no value") + ")");
    };
    // str_sub's positions: negative counts from the end, clipped to [1, len]
    auto posrelat = [](double pos, size_t len) -> long {
        long p = (long)pos;
        if (p >= 0) return p;
        if ((size_t)-p > len) return 0;
        return (long)len + p + 1;
    };
    register_builtin("string.len", [argstr](Interp &, const Values &a, Values &r) { r.push_back(Value::number((double)argstr(a, 0, "len").size())); });
    register_builtin("string.sub", [argstr, posrelat](Interp &, const Values &a, Values &r) {
        const std::string s = argstr(a, 0, "sub");
        long i = posrelat(argnum(a, 1, "sub"), s.size()), j = posrelat(a.size() > 2 && a[2].t != Value::NIL ? argnum(a, 2, "sub") : -1, s.size());
        if (i < 1) i = 1;
        if (j > (long)s.size()) j = (long)s.size();
        r.push_back(Value::string(i <= j ? s.substr((size_t)i - 1, (size_t)(j - i + 1)) : std::string()));
    });
    register_builtin("string.upper", [argstr](Interp &, const Values &a, Values &r) {
        std::string s = argstr(a, 0, "upper");
        for (char &c : s) c = (char)toupper((unsigned char)c);
        r.push_back(Value::string(s));
    });
    register_builtin("string.lower", [argstr](Interp &, const Values &a, Values &r) {
        std::string s = argstr(a, 0, "lower");
        for (char &c : s) c = (char)tolower((unsigned char)c);
        r.push_back(Value::string(s));
    });
    register_builtin("string.reverse", [argstr](Interp &, const Values &a, Values &r) {
        std::string s = argstr(a, 0, "reverse");
        r.push_back(Value::string(std::string(s.rbegin(), s.rend())));
    });
    register_builtin("string.rep", [argstr](Interp &, const Values &a, Values &r) {
        const std::string s = argstr(a, 0, "rep"), sep = a.size() > 2 ? argstr(a, 2, "rep") : std::string();
        const long n = (long)argnum(a, 1, "rep");
        std::string out;
        if (n > 0 && (s.size() + sep.size()) * (size_t)n > (size_t)1 << 24) throw LuaError("resulting string too large");
        for (long i = 0; i < n; ++i) { out += s; if (i + 1 < n) out += sep; }
        r.push_back(Value::string(out));
    });
    register_builtin("string.byte", [argstr, posrelat](Interp &, const Values &a, Values &r) {
        const std::string s = argstr(a, 0, "byte");
        long i = posrelat(a.size() > 1 && a[1].t != Value::NIL ? argnum(a, 1, "byte") : 1, s.size());
        long j = posrelat(a.size() > 2 && a[2].t != Value::NIL ? argnum(a, 2, "byte") : (double)i, s.size());
        if (i < 1) i = 1;
        if (j > (long)s.size()) j = (long)s.size();
        for (long k = i; k <= j; ++k) r.push_back(Value::number((double)(unsigned char)s[(size_t)k - 1]));
    });
    register_builtin("string.char", [](Interp &, const Values &a, Values &r) {
        std::string out;
        for (size_t i = 0; i < a.size(); ++i) {
            const double c = argnum(a, i, "char");
            if (c < 0 || c > 255 || c != std::floor(c)) throw LuaError("bad argument #" + std::to_string(i + 1) + " to 'char' (value out of range)");
            out += (char)(unsigned char)c;
        }
        r.push_back(Value::string(out));
    });
    // string.find / match share str_find_aux: init, plain search when asked for or when the pattern has no magic characters
    auto find_or_match = [argstr, posrelat](const Values &a, Values &r, bool find) {
        const char *fn = find ? "find" : "match";
        const std::string s = argstr(a, 0, fn), pat = argstr(a, 1, fn);
        long init = posrelat(a.size() > 2 && a[2].t != Value::NIL ? argnum(a, 2, fn) : 1, s.size());
        if (init < 1) init = 1;
        if ((size_t)init > s.size() + 1) { r.push_back(Value()); return; }
        if (find && ((a.size() > 3 && a[3].truthy()) || pat.find_first_of("^$*+?.([%-") == std::string::npos)) {
            const size_t at = s.find(pat, (size_t)init - 1);
            if (at == std::string::npos) { r.push_back(Value()); return; }
            r.push_back(Value::number((double)at + 1));
            r.push_back(Value::number((double)(at + pat.size())));
            return;
        }
        PatternMatch m(s, pat);
        size_t b = 0, e = 0;
        if (!pattern_search(m, (size_t)init - 1, &b, &e)) { r.push_back(Value()); return; }
        if (find) {
            r.push_back(Value::number((double)b + 1));
            r.push_back(Value::number((double)e));
            m.push_captures(r, b, e, false);
        } else m.push_captures(r, b, e, true);
    };
    register_builtin("string.find", [find_or_match](Interp &, const Values &a, Values &r) { find_or_match(a, r, true); });
    register_builtin("string.match", [find_or_match](Interp &, const Values &a, Values &r) { find_or_match(a, r, false); });
    register_builtin("string.gmatch", [argstr](Interp &I, const Values &a, Values &r) {
        auto subject = std::make_shared<std::string>(argstr(a, 0, "gmatch"));
        auto pattern = std::make_shared<std::string>(argstr(a, 1, "gmatch"));
        auto pos = std::make_shared<size_t>(0);
        Value iter;
        iter.t = Value::BUILTIN;
        auto bp = std::make_shared<Builtin>();
        bp->name = "gmatch iterator";
        bp->fn = [subject, pattern, pos](Interp &, const Values &, Values &out) {
            while (*pos <= subject->size()) {
                PatternMatch m(*subject, *pattern);
                m.caps.clear();
                const size_t e = m.match(*pos, 0);                 // (gmatch does not anchor: a '^' is a literal here, lstrlib.c)
                if (e != PatternMatch::NO) {
                    const size_t b = *pos;
                    *pos = e == b ? e + 1 : e;                      // an empty match moves on by one
                    m.push_captures(out, b, e, true);
                    return;
                }
                ++*pos;
            }
            out.push_back(Value());
        };
        iter.p = std::move(bp);
        (void)I;
        r.push_back(iter);
    });
    register_builtin("string.gsub", [argstr](Interp &I, const Values &a, Values &r) {
        const std::string s = argstr(a, 0, "gsub"), pat = argstr(a, 1, "gsub");
        if (a.size() < 3 || !(a[2].t == Value::STR || a[2].t == Value::NUM || a[2].t == Value::TABLE || a[2].is_function()))
            throw LuaError("bad argument #3 to 'gsub' (string/function/table expected)");
        const Value &repl = a[2];
        const double max_n = a.size() > 3 && a[3].t != Value::NIL ? argnum(a, 3, "gsub") : (double)s.size() + 1;
        const bool anchored = !pat.empty() && pat[0] == '^';
        std::string out;
        size_t si = 0, n = 0;
        while ((double)n < max_n) {
            PatternMatch m(s, pat);
            const size_t e = m.match(si, anchored ? 1 : 0);
            if (e != PatternMatch::NO) {
                ++n;
                const Value whole = Value::string(s.substr(si, e - si));
                Value with;
                if (repl.t == Value::STR || repl.t == Value::NUM) {
                    const std::string rs = I.tostring(repl);
                    std::string built;
                    for (size_t k = 0; k < rs.size(); ++k) {
                        if (rs[k] != '%') { built += rs[k]; continue; }
                        if (++k >= rs.size()) throw LuaError("invalid use of '%' in replacement string");
                        if (rs[k] == '%') built += '%';
                        else if (rs[k] == '0') built += whole.str();
                        else if (isdigit((unsigned char)rs[k])) built += I.tostring(m.capture((size_t)(rs[k] - '1'), si, e));
                        else throw LuaError("invalid use of '%' in replacement string");
                    }
                    with = Value::string(built);
                } else {
                    const Value key = m.capture(0, si, e);
                    if (repl.t == Value::TABLE) with = repl.tab()->get(key);
                    else {
                        Values args;
                        m.push_captures(args, si, e, true);
                        Values res = I.call(repl, args);
                        with = res.empty() ? Value() : res[0];
                    }
                }
                if (!with.truthy()) out += whole.str();                            // false / nil: keep the match
                else if (with.t == Value::STR || with.t == Value::NUM) out += I.tostring(with);
                else throw LuaError(std::string("invalid replacement value (a ") + with.type_name() + ")");
            }
            if (e != PatternMatch::NO && e > si) si = e;
            else if (si < s.size()) out += s[si++];
            else break;
            if (anchored) break;
        }
        if (si < s.size()) out += s.substr(si);
        r.push_back(Value::string(out));
        r.push_back(Value::number((double)n));
    });
    // str_format: one C conversion per directive, flags / width / precision as given (lstrlib.c:scanformat)
    register_builtin("string.format", [argstr](Interp &I, const Values &a, Values &r) {
        const std::string fmt = argstr(a, 0, "format");
        std::string out;
        size_t arg = 0;
        for (size_t i = 0; i < fmt.size(); ++i) {
            if (fmt[i] != '%') { out += fmt[i]; continue; }
            if (++i >= fmt.size()) throw LuaError("invalid option '%' to 'format'");
            if (fmt[i] == '%') { out += '%'; continue; }
            std::string spec = "%";
            while (i < fmt.size() && strchr("-+ #0", fmt[i])) spec += fmt[i++];
            while (i < fmt.size() && isdigit((unsigned char)fmt[i])) spec += fmt[i++];
            if (i < fmt.size() && fmt[i] == '.') { spec += fmt[i++]; while (i < fmt.size() && isdigit((unsigned char)fmt[i])) spec += fmt[i++]; }
            if (i >= fmt.size() || spec.size() > 20) throw LuaError("invalid format (width or precision too long)");
            const char conv = fmt[i];
            ++arg;
            char buf[512];
            switch (conv) {
            case 'c': out += (char)(int)argnum(a, arg, "format"); break;
            case 'd': case 'i':
                snprintf(buf, sizeof buf, (spec + "lld").c_str(), (long long)argnum(a, arg, "format"));
                out += buf;
                break;
            case 'o': case 'u': case 'x': case 'X':
                snprintf(buf, sizeof buf, (spec + "ll" + conv).c_str(), (unsigned long long)(long long)argnum(a, arg, "format"));
                out += buf;
                break;
            case 'e': case 'E': case 'f': case 'g': case 'G': case 'a': case 'A':
                snprintf(buf, sizeof buf, (spec + conv).c_str(), argnum(a, arg, "format"));
                out += buf;
                break;
            case 'q': {
                const std::string v = argstr(a, arg, "format");
                out += '"';
                for (char c : v) {
                    if (c == '"' || c == '\\' || c == '\n') { out += '\\'; out += c; }
                    else if (c == '\r') out += "\\r";
                    else if (c == '\0') out += "\\0";
                    else out += c;
                }
                out += '"';
                break;
            }
            case 's': {
                if (arg >= a.size()) throw LuaError("bad argument #" + std::to_string(arg + 1) + " to 'format' (no value)");
                const std::string v = I.tostring(a[arg]);                      // luaL_tolstring: any value
                if (spec == "%") out += v;
                else {
                    std::vector<char> big(v.size() + 64);
                    snprintf(big.data(), big.size(), (spec + "s").c_str(), v.c_str());
                    out += big.data();
                }
                break;
            }
            default: throw LuaError(std::string("invalid option '%") + conv + "' to 'format'");
            }
        }
        r.push_back(Value::string(out));
    });
    // ---- more of the table library ----------------------------------------------------------------------------------------------
    register_builtin("table.concat", [](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'concat' (table expected)");
        const Table &t = *a[0].tab();
        const std::string sep = a.size() > 1 && a[1].t != Value::NIL ? I.tostring(a[1]) : std::string();
        const long i0 = a.size() > 2 && a[2].t != Value::NIL ? (long)argnum(a, 2, "concat") : 1;
        const long i1 = a.size() > 3 && a[3].t != Value::NIL ? (long)argnum(a, 3, "concat") : (long)t.length();
        std::string out;
        for (long i = i0; i <= i1; ++i) {
            const Value v = t.get(Value::number((double)i));
            if (v.t != Value::STR && v.t != Value::NUM) throw LuaError("invalid value (at index " + std::to_string(i) + ") in table for 'concat'");
            out += I.tostring(v);
            if (i < i1) out += sep;
        }
        r.push_back(Value::string(out));
    });
    register_builtin("table.remove", [](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'remove' (table expected)");
        Table &t = *a[0].tab();
        const size_t n = t.length();
        size_t pos = a.size() > 1 && a[1].t != Value::NIL ? (size_t)argnum(a, 1, "remove") : n;
        if (n == 0 && (a.size() < 2 || pos == 0 || pos == n)) { r.push_back(Value()); return; }
        if (pos < 1 || pos > n + 1) throw LuaError("bad argument #2 to 'remove' (position out of bounds)");
        if (pos == n + 1) { r.push_back(Value()); return; }
        r.push_back(t.arr[pos - 1]);
        t.arr.erase(t.arr.begin() + (long)(pos - 1));
    });
    register_builtin("table.sort", [](Interp &I, const Values &a, Values &) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'sort' (table expected)");
        Table &t = *a[0].tab();
        const bool custom = a.size() > 1 && a[1].t != Value::NIL;
        if (custom && !a[1].is_function()) throw LuaError("bad argument #2 to 'sort' (function expected)");
        const Value cmp = custom ? a[1] : Value();
        auto less = [&](const Value &x, const Value &y) {
            if (custom) { Values r = I.call(cmp, Values{x, y}); return !r.empty() && r[0].truthy(); }
            if (x.t == Value::NUM && y.t == Value::NUM) return x.n < y.n;
            if (x.t == Value::STR && y.t == Value::STR) return x.str() < y.str();
            throw LuaError(std::string("attempt to compare ") + x.type_name() + " with " + y.type_name());
        };
        // insertion by binary search: well defined for any comparison function, and lens scripts sort a handful of values
        std::vector<Value> out;
        for (const Value &v : t.arr) {
            size_t lo = 0, hi = out.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (less(v, out[mid])) hi = mid; else lo = mid + 1; }
            out.insert(out.begin() + (long)lo, v);
        }
        t.arr.swap(out);
    });
    // ---- the rest of lmathlib.c that is plain arithmetic ------------------------------------------------------------------------
    register_builtin("math.ldexp", [](Interp &, const Values &a, Values &r) { r.push_back(Value::number(std::ldexp(argnum(a, 0, "ldexp"), (int)argnum(a, 1, "ldexp")))); });
    register_builtin("math.frexp", [](Interp &, const Values &a, Values &r) {
        int e = 0;
        r.push_back(Value::number(std::frexp(argnum(a, 0, "frexp"), &e)));
        r.push_back(Value::number((double)e));
    });
    // math.random / math.randomseed as lmathlib.c has them, over the C library's rand() - the sequence the reference's VM would draw
    // on this platform.  Host only: the GPU build rejects them (every pixel would need the draws of all pixels before it).
    register_builtin("math.random", [](Interp &, const Values &a, Values &r) {
        const double x = (double)(rand() % RAND_MAX) / (double)RAND_MAX;
        if (a.empty()) { r.push_back(Value::number(x)); return; }
        const double lo = a.size() > 1 ? argnum(a, 0, "random") : 1.0, hi = argnum(a, a.size() > 1 ? 1 : 0, "random");
        if (lo > hi) throw LuaError("bad argument #" + std::to_string(a.size() > 1 ? 2 : 1) + " to 'random' (interval is empty)");
        r.push_back(Value::number(std::floor(x * (hi - lo + 1)) + lo));
    });
    register_builtin("math.randomseed", [](Interp &, const Values &a, Values &) { srand((unsigned)argnum(a, 0, "randomseed")); (void)rand(); });
    register_builtin("next", [](Interp &, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'next' (table expected)");
        const Table &t = *a[0].tab();
        const Value k = a.size() > 1 ? a[1] : Value();
        // order: array part, numeric hash, string hash
        size_t ai = 0;
        bool after_arr = false;
        if (k.t == Value::NIL) ai = 0;
        else if (k.t == Value::NUM && k.n >= 1 && k.n <= (double)t.arr.size() && k.n == std::floor(k.n)) ai = (size_t)k.n;
        else after_arr = true;
        if (!after_arr) {
            while (ai < t.arr.size() && t.arr[ai].t == Value::NIL) ++ai;          // (holes are not entries)
            if (ai < t.arr.size()) { r.push_back(Value::number((double)ai + 1)); r.push_back(t.arr[ai]); return; }
            if (!t.nhash.empty()) { r.push_back(Value::number(t.nhash.begin()->first)); r.push_back(t.nhash.begin()->second); return; }
            if (!t.shash.empty()) { r.push_back(Value::string(t.shash.begin()->first)); r.push_back(t.shash.begin()->second); return; }
            r.push_back(Value());
            return;
        }
        if (k.t == Value::NUM) {
            auto it = t.nhash.find(k.n);
            if (it != t.nhash.end() && ++it != t.nhash.end()) { r.push_back(Value::number(it->first)); r.push_back(it->second); return; }
            if (!t.shash.empty()) { r.push_back(Value::string(t.shash.begin()->first)); r.push_back(t.shash.begin()->second); return; }
            r.push_back(Value());
            return;
        }
        if (k.t == Value::STR) {
            auto it = t.shash.find(k.str());
            if (it != t.shash.end() && ++it != t.shash.end()) { r.push_back(Value::string(it->first)); r.push_back(it->second); return; }
        }
        r.push_back(Value());
    });
    register_builtin("pairs", [](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'pairs' (table expected)");
        r.push_back(I.get_global("next"));
        r.push_back(a[0]);
        r.push_back(Value());
    });
    register_builtin("ipairs", [](Interp &I, const Values &a, Values &r) {
        if (a.empty() || a[0].t != Value::TABLE) throw LuaError("bad argument #1 to 'ipairs' (table expected)");
        Value it;
        it.t = Value::BUILTIN;
        auto itb = std::make_shared<Builtin>();
        itb->name = "ipairs_iter";
        itb->fn = [](Interp &, const Values &x, Values &out) {
            double i = x[1].n + 1;
            Value v = x[0].tab()->get(Value::number(i));
            if (v.t == Value::NIL) { out.push_back(Value()); return; }
            out.push_back(Value::number(i));
            out.push_back(v);
        };
        (void)I;
        it.p = std::move(itb);
        r.push_back(it);
        r.push_back(a[0]);
        r.push_back(Value::number(0));
    });
}

}  // namespace bklua
