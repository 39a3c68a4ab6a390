// ===========================================================================
// plo_host.hpp -- dependency-free C++17 host substrate of the drop-in:
// what the reference gets from Givaro/LinBox for this path, and no more.
//   * fields Z_p (canonical residues, integer order) and Q (exact small rationals)
//   * sparse row matrix, SMS reader/writer (reference README.md:71-77)
//   * SLP parser/evaluator -> matrix (what src/SLPchecker.cpp:22-105 verifies)
//   * `replay`: text-emitting Optimizer() for ONE seed -- the host replays the
//     seed the GPU search returned (include/plinopt_optimize.inl:616-631) and
//     also serves the rational (Q) path of bin/optimizer, which the reference
//     runs on the CPU and BASELINE config[0] keeps there ("plumbing, no GPU").
// This is product code; it never includes or links anything under oracle/.
// ===========================================================================
#ifndef PLO_HOST_HPP
#define PLO_HOST_HPP

#include <algorithm>
#include <sys/wait.h>
#include <unistd.h>
#include <signal.h>
#include <cerrno>
#include <cstring>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace plo {

// ------------------------------------------------------------------ rationals
// 128-bit numerator and denominator, overflow-checked (round 3; rounds 1-2: 64 bits).  The reference's Givaro::Rational is arbitrary
// precision; the fixtures with decimal coefficients (e.g. 2x2x2_7_DPS-intermediate-12.0695) overflow 64 bits after a few merges of
// the in-place programs, not 128.  An overflow is an exception, never a wrong value.
struct Rat {
    __int128 n = 0, d = 1;
    Rat() = default;
    Rat(int64_t nn, int64_t dd = 1) { set((__int128)nn, (__int128)dd); }
    static __int128 gcd(__int128 a, __int128 b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { __int128 t = a % b; a = b; b = t; } return a; }
    static __int128 mulck(__int128 a, __int128 b) { __int128 r; if (__builtin_mul_overflow(a, b, &r)) throw std::overflow_error("rational overflow (128-bit)"); return r; }
    static __int128 addck(__int128 a, __int128 b) { __int128 r; if (__builtin_add_overflow(a, b, &r)) throw std::overflow_error("rational overflow (128-bit)"); return r; }
    void set(__int128 nn, __int128 dd) {
        if (dd == 0) throw std::domain_error("rational with zero denominator");
        if (dd < 0) { nn = -nn; dd = -dd; }
        const __int128 g = gcd(nn, dd);
        if (g > 1) { nn /= g; dd /= g; }
        n = nn; d = dd;
    }
    static Rat make(__int128 nn, __int128 dd) { Rat r; r.set(nn, dd); return r; }
    // a.n/a.d * b.n/b.d and a.n/a.d + b.n/b.d with the common factors taken out before the products
    static Rat mul(const Rat &a, const Rat &b) { const __int128 g1 = gcd(a.n, b.d), g2 = gcd(b.n, a.d); return make(mulck(a.n / (g1 ? g1 : 1), b.n / (g2 ? g2 : 1)), mulck(a.d / (g2 ? g2 : 1), b.d / (g1 ? g1 : 1))); }
    static Rat add(const Rat &a, const Rat &b) { const __int128 g = gcd(a.d, b.d), bd = b.d / g, ad = a.d / g; return make(addck(mulck(a.n, bd), mulck(b.n, ad)), mulck(a.d, bd)); }
    bool operator==(const Rat &o) const { return n == o.n && d == o.d; }
    bool operator!=(const Rat &o) const { return !(*this == o); }
    bool operator<(const Rat &o) const { Rat m; m.n = -o.n; m.d = o.d; return add(*this, m).n < 0; }
};
inline std::ostream &operator<<(std::ostream &os, __int128 v) {
    char t[48]; int k = 47; t[k] = 0; const bool neg = v < 0; unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    do { t[--k] = (char)('0' + (int)(u % 10)); u /= 10; } while (u);
    if (neg) t[--k] = '-';
    return os << (t + k);
}
inline std::ostream &operator<<(std::ostream &os, const Rat &r) { os << r.n; if (r.d != 1) os << '/' << r.d; return os; }

inline Rat parse_rat(const std::string &s) {
    auto sl = s.find('/');
    if (sl == std::string::npos) return Rat(std::stoll(s));
    return Rat(std::stoll(s.substr(0, sl)), std::stoll(s.substr(sl + 1)));
}

// --------------------------------------------------------------------- fields
// Z_p with Givaro::Modular<Integer> conventions:
// canonical residues in [0,p), operator< is integer order, isMOne(e) <=> e==p-1,
// Fabs/Fsign as include/plinopt_library.h:209-225.
struct ZpField {
    using Elt = uint32_t;
    uint32_t p;
    explicit ZpField(uint32_t pp) : p(pp) {}
    Elt zero() const { return 0; }
    Elt one() const { return 1u % p; }
    Elt mone() const { return p - 1; }
    Elt mul(Elt a, Elt b) const { return (Elt)((uint64_t)a * b % p); }
    Elt add(Elt a, Elt b) const { return (Elt)(((uint64_t)a + b) % p); }
    Elt neg(Elt a) const { return a ? p - a : 0; }
    Elt inv(Elt a) const {
        int64_t t = 0, nt = 1, r = p, nr = a % p;
        while (nr) { int64_t q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; }
        if (r != 1) throw std::domain_error("element not invertible mod p");
        if (t < 0) t += p;
        return (Elt)t;
    }
    Elt div(Elt a, Elt b) const { return mul(a, inv(b)); }
    bool isZero(Elt a) const { return a == 0; }
    bool isOne(Elt a) const { return a == one(); }
    bool isMOne(Elt a) const { return a == mone(); }
    bool less(Elt a, Elt b) const { return a < b; }
    Elt abs(Elt e) const { Elt a = neg(e); return a < e ? a : e; }
    int sign(Elt e) const { if (!e) return 0; return neg(e) < e ? -1 : 1; }
    Elt fromRat(const Rat &r) const {
        int64_t nn = r.n % (int64_t)p; if (nn < 0) nn += p;
        Elt dd = (Elt)(r.d % (int64_t)p);
        if (dd == 0) throw std::domain_error("denominator vanishes mod p");
        return div((Elt)nn, dd);
    }
    Elt fromInt(int64_t v) const { int64_t x = v % (int64_t)p; if (x < 0) x += p; return (Elt)x; }
    void write(std::ostream &os, Elt e) const { os << e; }
    // printmulorjustdiv, generic: include/plinopt_library.inl:348-358
    void print_mul(std::ostream &os, char c, size_t i, Elt e, size_t &nbmul) const {
        os << c << i;
        if (!(isOne(e) || isMOne(e))) { ++nbmul; os << '*' << e; }
    }
    std::string name() const { return "Z/" + std::to_string(p) + "Z"; }
};

// Z/pZ for 2^31 <= p < 2^62 (the reference's field is Givaro::Modular<Integer>, src/optimizer.cpp:131: any modulus).  Host loops only: the
// HIP kernels keep 31-bit residues (44-bit pair keys); the tools fall back to the host engines for these moduli and say so.
struct Zp64Field {
    using Elt = uint64_t;
    uint64_t p;
    explicit Zp64Field(uint64_t pp) : p(pp) {}
    Elt zero() const { return 0; }
    Elt one() const { return 1u % p; }
    Elt mone() const { return p - 1; }
    Elt mul(Elt a, Elt b) const { return (Elt)((unsigned __int128)a * b % p); }
    Elt add(Elt a, Elt b) const { return (Elt)(((unsigned __int128)a + b) % p); }
    Elt neg(Elt a) const { return a ? p - a : 0; }
    Elt inv(Elt a) const {
        __int128 t = 0, nt = 1, r = p, nr = a % p;
        while (nr) { __int128 q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; }
        if (r != 1) throw std::domain_error("element not invertible mod p");
        if (t < 0) t += p;
        return (Elt)t;
    }
    Elt div(Elt a, Elt b) const { return mul(a, inv(b)); }
    bool isZero(Elt a) const { return a == 0; }
    bool isOne(Elt a) const { return a == one(); }
    bool isMOne(Elt a) const { return a == mone(); }
    bool less(Elt a, Elt b) const { return a < b; }
    Elt abs(Elt e) const { Elt a = neg(e); return a < e ? a : e; }
    int sign(Elt e) const { if (!e) return 0; return neg(e) < e ? -1 : 1; }
    Elt fromRat(const Rat &r) const {
        __int128 nn = (__int128)r.n % (__int128)p; if (nn < 0) nn += p;
        Elt dd = (Elt)((__int128)r.d % (__int128)p);
        if (dd == 0) throw std::domain_error("denominator vanishes mod p");
        return div((Elt)nn, dd);
    }
    Elt fromInt(int64_t v) const { __int128 x = (__int128)v % (__int128)p; if (x < 0) x += p; return (Elt)x; }
    void write(std::ostream &os, Elt e) const { os << e; }
    void print_mul(std::ostream &os, char c, size_t i, Elt e, size_t &nbmul) const {
        os << c << i;
        if (!(isOne(e) || isMOne(e))) { ++nbmul; os << '*' << e; }
    }
    std::string name() const { return "Z/" + std::to_string(p) + "Z"; }
};

struct QField {
    using Elt = Rat;
    Elt zero() const { return Rat(0); }
    Elt one() const { return Rat(1); }
    Elt mone() const { return Rat(-1); }
    Elt mul(const Elt &a, const Elt &b) const { return Rat::mul(a, b); }
    Elt add(const Elt &a, const Elt &b) const { return Rat::add(a, b); }
    Elt neg(const Elt &a) const { return Rat::make(-(__int128)a.n, a.d); }
    Elt inv(const Elt &a) const { return Rat::make(a.d, a.n); }
    Elt div(const Elt &a, const Elt &b) const { return Rat::mul(a, Rat::make(b.d, b.n)); }
    bool isZero(const Elt &a) const { return a.n == 0; }
    bool isOne(const Elt &a) const { return a.n == 1 && a.d == 1; }
    bool isMOne(const Elt &a) const { return a.n == -1 && a.d == 1; }
    bool less(const Elt &a, const Elt &b) const { return a < b; }
    Elt abs(const Elt &e) const { return e.n < 0 ? neg(e) : e; }
    int sign(const Elt &e) const { return e.n > 0 ? 1 : (e.n < 0 ? -1 : 0); }
    Elt fromRat(const Rat &r) const { return r; }
    Elt fromInt(int64_t v) const { return Rat(v); }
    void write(std::ostream &os, const Elt &e) const { os << e; }
    // printmulorjustdiv, rational specialisation: include/plinopt_library.inl:360-374
    void print_mul(std::ostream &os, char c, size_t i, const Elt &r, size_t &nbmul) const {
        os << c << i;
        if (!isOne(r)) { ++nbmul; if (r.n == 1) os << '/' << r.d; else os << '*' << r; }
    }
    std::string name() const { return "Q"; }
};

template <class F> inline bool absOne(const F &f, const typename F::Elt &e) { return f.isOne(e) || f.isMOne(e); }

// --------------------------------------------------------------------- matrix
template <class E> struct SparseMat {
    using Row = std::vector<std::pair<size_t, E>>;
    std::vector<Row> rows;
    size_t ncols = 0;
    SparseMat() = default;
    SparseMat(size_t m, size_t n) : rows(m), ncols(n) {}
    size_t rowdim() const { return rows.size(); }
    size_t coldim() const { return ncols; }
    size_t nnz() const { size_t s = 0; for (auto &r : rows) s += r.size(); return s; }
};
using QMat = SparseMat<Rat>;

template <class E> SparseMat<E> transpose(const SparseMat<E> &A) {   // plinopt_library.inl:18-24
    SparseMat<E> T(A.coldim(), A.rowdim());
    for (size_t i = 0; i < A.rowdim(); ++i) for (auto &e : A.rows[i]) T.rows[e.first].emplace_back(i, e.second);
    return T;
}
// rebind<Field>::other(M,F): rational -> field image, entries that vanish are dropped (Appendix B)
template <class F> SparseMat<typename F::Elt> rebind(const QMat &M, const F &f) {
    SparseMat<typename F::Elt> R(M.rowdim(), M.coldim());
    for (size_t i = 0; i < M.rowdim(); ++i)
        for (auto &e : M.rows[i]) { auto v = f.fromRat(e.second); if (!f.isZero(v)) R.rows[i].emplace_back(e.first, v); }
    return R;
}
// naiveOps, plinopt_library.inl:227-235
template <class F, class E> std::pair<size_t, size_t> naive_ops(const F &f, const SparseMat<E> &M) {
    size_t a = 0, mu = 0;
    for (auto &r : M.rows) { if (r.size() > 1) a += r.size() - 1; for (auto &e : r) if (!absOne(f, e.second)) ++mu; }
    return {a, mu};
}

// SMS: header `m n R|M`, 1-based triples, `0 0 0` terminator, `#`/`%` comments, ints or a/b
inline QMat read_sms(std::istream &in) {
    std::string line; bool header = false; QMat M;
    std::vector<std::tuple<size_t, size_t, Rat>> ent;
    size_t m = 0, n = 0;
    while (std::getline(in, line)) {
        size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos || line[b] == '#' || line[b] == '%') continue;
        std::istringstream ls(line);
        if (!header) { std::string ty; if (!(ls >> m >> n)) throw std::runtime_error("SMS: bad header"); ls >> ty; header = true; continue; }
        long long i, j; std::string v;
        if (!(ls >> i >> j >> v)) throw std::runtime_error("SMS: bad triple: " + line);
        if (i == 0 && j == 0) break;
        if (i < 1 || j < 1 || (size_t)i > m || (size_t)j > n) throw std::runtime_error("SMS: index out of range: " + line);
        Rat r = parse_rat(v);
        if (r.n != 0) ent.emplace_back((size_t)i - 1, (size_t)j - 1, r);
    }
    if (!header) throw std::runtime_error("SMS: empty input");
    M = QMat(m, n);
    for (auto &t : ent) {
        auto &row = M.rows[std::get<0>(t)];
        auto it = std::lower_bound(row.begin(), row.end(), std::get<1>(t), [](const std::pair<size_t, Rat> &e, size_t c) { return e.first < c; });
        if (it != row.end() && it->first == std::get<1>(t)) it->second = std::get<2>(t); else row.insert(it, {std::get<1>(t), std::get<2>(t)});
    }
    return M;
}
template <class F, class E> void write_sms(std::ostream &os, const F &f, const SparseMat<E> &M, char ty = 'R') {
    os << M.rowdim() << ' ' << M.coldim() << ' ' << ty << '\n';
    for (size_t i = 0; i < M.rowdim(); ++i) for (auto &e : M.rows[i]) { os << i + 1 << ' ' << e.first + 1 << ' '; f.write(os, e.second); os << '\n'; }
    os << "0 0 0\n";
}

// ------------------------------------------------------------------------ SLP
// Straight-line programs `lhs:=expr;` over + - * / ( ), naturals and a/b
// constants (reference README.md:83-92).  evaluate(): matrix of the outputs
// `o#` in terms of the inputs `i#` (matrixBuilder, plinopt_programs.inl:1458-1608).
template <class F> class SlpEval {
    using E = typename F::Elt;
    using Lin = std::vector<std::pair<long, E>>;       // sorted by index; index -1 = constant term
    const F &f;
    std::map<std::string, Lin> vars;
    std::vector<std::string> tk; size_t pos = 0;

    Lin axpy(const Lin &x, const Lin &y, bool minus) const {
        Lin o; o.reserve(x.size() + y.size());
        size_t a = 0, b = 0;
        while (a < x.size() || b < y.size()) {
            if (b == y.size() || (a < x.size() && x[a].first < y[b].first)) o.push_back(x[a++]);
            else if (a == x.size() || y[b].first < x[a].first) { o.emplace_back(y[b].first, minus ? f.neg(y[b].second) : y[b].second); ++b; }
            else { E v = f.add(x[a].second, minus ? f.neg(y[b].second) : y[b].second); if (!f.isZero(v)) o.emplace_back(x[a].first, v); ++a; ++b; }
        }
        return o;
    }
    Lin scale(const Lin &x, const E &c) const { Lin o; for (auto &e : x) { E v = f.mul(e.second, c); if (!f.isZero(v)) o.emplace_back(e.first, v); } return o; }
    static bool isconst(const Lin &x) { return x.empty() || (x.size() == 1 && x[0].first == -1); }
    E constof(const Lin &x) const { return x.empty() ? f.zero() : x[0].second; }
    const std::string &peek() const { static const std::string none; return pos < tk.size() ? tk[pos] : none; }
    Lin expr() {
        bool neg = false;
        if (peek() == "+" || peek() == "-") { neg = tk[pos++] == "-"; }
        Lin acc = term(); if (neg) acc = scale(acc, f.mone());
        while (peek() == "+" || peek() == "-") { bool mi = tk[pos++] == "-"; acc = axpy(acc, term(), mi); }
        return acc;
    }
    Lin term() {
        Lin acc = factor();
        while (peek() == "*" || peek() == "/") {
            bool dv = tk[pos++] == "/"; Lin r = factor();
            if (dv) { if (!isconst(r)) throw std::runtime_error("SLP: division by a non-constant"); acc = scale(acc, f.inv(constof(r))); }
            else if (isconst(r)) acc = scale(acc, constof(r));
            else if (isconst(acc)) acc = scale(r, constof(acc));
            else throw std::runtime_error("SLP: non-linear product");
        }
        return acc;
    }
    Lin factor() {
        if (pos >= tk.size()) throw std::runtime_error("SLP: unexpected end of line");
        std::string t = tk[pos++];
        if (t == "(") { Lin v = expr(); if (peek() != ")") throw std::runtime_error("SLP: missing )"); ++pos; return v; }
        if (t == "-") return scale(factor(), f.mone());
        if (isdigit((unsigned char)t[0])) { E c = f.fromInt(std::stoll(t)); Lin v; if (!f.isZero(c)) v.emplace_back(-1, c); return v; }
        auto it = vars.find(t);
        if (it != vars.end()) return it->second;
        if (t[0] == 'i' && t.size() > 1 && isdigit((unsigned char)t[1])) { Lin v; v.emplace_back(std::stol(t.substr(1)), f.one()); return v; }
        throw std::runtime_error("SLP: undefined variable " + t);
    }
public:
    explicit SlpEval(const F &ff) : f(ff) {}
    static std::vector<std::string> tokenize(const std::string &line) {
        std::vector<std::string> out; size_t i = 0;
        while (i < line.size()) {
            char c = line[i];
            if (isspace((unsigned char)c)) { ++i; continue; }
            if (c == ':' && i + 1 < line.size() && line[i + 1] == '=') { out.emplace_back(":="); i += 2; continue; }
            if (isalpha((unsigned char)c) || c == '_') { size_t j = i; while (j < line.size() && (isalnum((unsigned char)line[j]) || line[j] == '_')) ++j; out.push_back(line.substr(i, j - i)); i = j; continue; }
            if (isdigit((unsigned char)c)) { size_t j = i; while (j < line.size() && isdigit((unsigned char)line[j])) ++j; out.push_back(line.substr(i, j - i)); i = j; continue; }
            out.emplace_back(1, c); ++i;
        }
        return out;
    }
    // returns (adds, muls) with lineOperations semantics (plinopt_programs.inl:116-133)
    std::pair<size_t, size_t> run(std::istream &in) {
        std::string line; size_t adds = 0, muls = 0;
        while (std::getline(in, line)) {
            auto h = line.find('#'); if (h != std::string::npos) line.resize(h);
            if (line.find(":=") == std::string::npos) continue;
            tk = tokenize(line); pos = 0;
            if (tk.size() < 3 || tk[1] != ":=") throw std::runtime_error("SLP: bad line: " + line);
            // op count: a sign right after := or ( is a negation; a/b with both natural is one constant
            bool negator = false;
            for (size_t k = 0; k < tk.size(); ++k) {
                const std::string &w = tk[k];
                bool ratl = w == "/" && k > 0 && k + 1 < tk.size() && isdigit((unsigned char)tk[k - 1][0]) && isdigit((unsigned char)tk[k + 1][0]);
                if ((w == "+" || w == "-") && !negator) ++adds;
                else if ((w == "*" || w == "/") && !ratl) ++muls;
                negator = (w == ":=" || w == "(");
            }
            std::string lhs = tk[0]; pos = 2;
            Lin v = expr();
            if (pos < tk.size() && peek() != ";") throw std::runtime_error("SLP: trailing tokens in: " + line);   // a missing final ; is tolerated, as by programParser
            vars[lhs] = std::move(v);
        }
        return {adds, muls};
    }
    // outputs `o#` as a matrix (rows = outputs, cols = inputs); dimensions cover what is used
    SparseMat<E> matrix(char outchar = 'o', size_t minrows = 0, size_t mincols = 0) const {
        size_t m = minrows, n = mincols;
        for (auto &kv : vars) if (kv.first[0] == outchar && kv.first.size() > 1 && isdigit((unsigned char)kv.first[1])) {
            m = std::max(m, (size_t)std::stoul(kv.first.substr(1)) + 1);
            for (auto &e : kv.second) { if (e.first < 0) throw std::runtime_error("SLP: constant term in output " + kv.first); n = std::max(n, (size_t)e.first + 1); }
        }
        SparseMat<E> A(m, n);
        for (auto &kv : vars) if (kv.first[0] == outchar && kv.first.size() > 1 && isdigit((unsigned char)kv.first[1])) {
            auto &row = A.rows[std::stoul(kv.first.substr(1))];
            for (auto &e : kv.second) row.emplace_back((size_t)e.first, e.second);
        }
        return A;
    }
};

template <class F, class E> bool same_matrix(const F &f, const SparseMat<E> &A, const SparseMat<E> &B) {
    size_t m = std::max(A.rowdim(), B.rowdim());
    static const typename SparseMat<E>::Row empty;
    for (size_t i = 0; i < m; ++i) {
        const auto &a = i < A.rowdim() ? A.rows[i] : empty; const auto &b = i < B.rowdim() ? B.rows[i] : empty;
        if (a.size() != b.size()) return false;
        for (size_t k = 0; k < a.size(); ++k) if (a[k].first != b[k].first || !(a[k].second == b[k].second)) return false;
    }
    (void)f; return true;
}

// ---------------------------------------------------------------------- replay
// Per-candidate random stream (include/plinopt_hip.h): GivRandom LCG seeded per candidate.
struct CandRng {
    uint32_t s;
    explicit CandRng(uint64_t seed) {
        uint64_t x = seed + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        s = 1u + (uint32_t)(x % 2147483646ull);
    }
    uint32_t next() { s = (uint32_t)((950706376ull * (uint64_t)s) % 2147483647ull); return s; }
};

// Text-emitting restatement of Optimizer() for one seed.  Structured around the
// same containers the reference uses (ordered map of pair triples) because the
// order of that map IS the tie-break order.
template <class F> class Replay {
    using E = typename F::Elt;
    using Mat = SparseMat<E>;
    struct Tri { size_t a, b; E r; };
    struct TriLess {
        const F *f;
        bool operator()(const Tri &x, const Tri &y) const {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            return f->less(x.r, y.r);
        }
    };
    struct Mult { size_t idx, col; E val; };
    const F &f; Mat M; CandRng own_rng; CandRng &rng; std::ostream &out;
public:
    // -E: the candidate is the schedule of index `erem` in RecSub's tree (mixed radix of its own path, include/plinopt_hip.h)
    bool enumerate = false; uint64_t erem = 0, eprod = 1;
    void set_schedule(uint64_t index) { enumerate = true; erem = index; eprod = 1; }
private:
    char ouv, tev, rav;
    std::vector<Mult> multiples;
    size_t nbadd = 0, nbmul = 0;

    bool eq(const Tri &x, const Tri &y) const { return x.a == y.a && x.b == y.b && x.r == y.r; }
    std::vector<Tri> listpairs(const typename Mat::Row &row) const {      // :30-41
        std::vector<Tri> v;
        for (size_t x = 0; x < row.size(); ++x) for (size_t y = x + 1; y < row.size(); ++y)
            v.push_back(Tri{row[x].first, row[y].first, f.div(row[y].second, row[x].second)});
        return v;
    }
    bool find_mult(size_t col, const E &v, size_t &idx) const {
        for (auto &mm : multiples) if (mm.col == col && mm.val == v) { idx = mm.idx; return true; }
        return false;
    }
    void rem_one_cse(const Tri &cse, std::vector<std::vector<Tri>> &AP, std::map<Tri, size_t, TriLess> &PM) {   // :60-194
        size_t lm = M.ncols, c0 = 0, c1 = 0;
        for (auto &row : M.rows) for (auto &e : row) {
            if (e.first == cse.a && absOne(f, e.second)) ++c0;
            if (e.first == cse.b && absOne(f, e.second)) ++c1;
        }
        Tri l = cse;
        if (c0 < c1) l = Tri{cse.b, cse.a, f.inv(cse.r)};
        for (size_t i = 0; i < M.rowdim(); ++i) {
            auto &ap = AP[i];
            if (std::find_if(ap.begin(), ap.end(), [&](const Tri &t) { return eq(t, cse); }) == ap.end()) continue;
            auto &row = M.rows[i]; E coeff = f.zero();
            for (auto it = row.begin(); it != row.end(); ++it) if (it->first == l.a) { coeff = it->second; row.erase(it); break; }
            for (auto it = row.begin(); it != row.end(); ++it) if (it->first == l.b) { row.erase(it); row.emplace_back(lm, coeff); break; }
            for (auto &t : ap) { auto it = PM.find(t); if (--it->second == 0) PM.erase(it); }
            std::vector<Tri> nr;
            for (auto &t : ap) if (t.a != l.a && t.b != l.a && t.a != l.b && t.b != l.b) nr.push_back(t);
            for (size_t k = 0; k + 1 < row.size(); ++k) nr.push_back(Tri{row[k].first, row.back().first, f.div(row.back().second, row[k].second)});
            for (auto &t : nr) PM[t]++;
            ap.swap(nr);
        }
        E asgs = f.abs(l.r); size_t rindex = lm;
        if (!absOne(f, asgs)) {
            if (!find_mult(l.b, asgs, rindex)) {
                out << rav << lm << ":="; f.print_mul(out, tev, l.b, asgs, nbmul); out << ";\n";
                multiples.push_back(Mult{lm, l.b, asgs});
            }
        }
        out << tev << lm << ":=" << tev << l.a << ((f.isMOne(asgs) || f.sign(l.r) < 0) ? '-' : '+');
        if (absOne(f, asgs)) out << tev << l.b; else out << rav << rindex;
        out << ";\n";
        M.ncols = lm + 1;
    }
    bool one_sub() {                                                                                    // :209-314
        std::vector<std::vector<Tri>> AP;
        for (auto &row : M.rows) AP.push_back(listpairs(row));
        std::map<Tri, size_t, TriLess> PM(TriLess{&f});
        for (auto &ap : AP) for (auto &t : ap) PM[t]++;
        if (PM.empty()) return false;
        bool good = false;
        while (!PM.empty()) {
            size_t maxfrq = 0; std::vector<Tri> mx;
            for (auto &kv : PM) {
                if (kv.second == maxfrq) mx.push_back(kv.first);
                if (kv.second > maxfrq) { maxfrq = kv.second; mx.assign(1, kv.first); }
            }
            if (maxfrq <= 1) return good;
            good = true;
            if (enumerate) { mx.clear(); for (auto &kv : PM) if (kv.second > 1) mx.push_back(kv.first); }      // RecSub :935-937: every pair of frequency > 1
            Tri cse = mx[0];
            if (mx.size() > 1) {
                if (enumerate) { cse = mx[erem % mx.size()]; erem /= mx.size(); eprod = eprod > UINT64_MAX / mx.size() ? UINT64_MAX : eprod * mx.size(); }
                else cse = mx[rng.next() % mx.size()];
            }
            ++nbadd;
            rem_one_cse(cse, AP, PM);
        }
        return true;
    }
    std::map<E, size_t, std::function<bool(const E &, const E &)>> histo(const typename Mat::Row &row) const {
        std::map<E, size_t, std::function<bool(const E &, const E &)>> h([this](const E &x, const E &y) { return f.less(x, y); });
        for (auto &e : row) h[f.abs(e.second)]++;
        return h;
    }
    void factor_out_columns(Mat &T, size_t j) {                                                          // :318-371
        if (T.rows[j].empty()) return;
        auto h = histo(T.rows[j]);
        for (auto &kv : h) {
            size_t m = T.rowdim();
            if (kv.second > 1 && !absOne(f, kv.first)) {
                size_t rindex = m;
                if (!find_mult(j, kv.first, rindex)) {
                    out << rav << m << ":="; f.print_mul(out, tev, j, kv.first, nbmul); out << ";\n";
                    multiples.push_back(Mult{m, j, kv.first});
                }
                out << tev << m << ":=" << rav << rindex << ";\n";
                T.rows.emplace_back(); ++m;
                for (size_t k = 0; k < kv.second; ++k) {
                    auto &row = T.rows[j];
                    for (auto it = row.begin(); it != row.end(); ++it) if (f.abs(it->second) == kv.first) {
                        T.rows[m - 1].emplace_back(it->first, f.sign(it->second) >= 0 ? f.one() : f.mone());
                        row.erase(it); break;
                    }
                }
            }
        }
    }
    void factor_out_rows(Mat &A, size_t i) {                                                             // :375-420
        if (A.rows[i].empty()) return;
        auto h = histo(A.rows[i]);
        size_t m = A.ncols;
        for (auto &kv : h) {
            if (kv.second > 1 && !absOne(f, kv.first)) {
                out << tev << m << ":="; ++m; A.ncols = m;
                auto &row = A.rows[i];
                row.emplace_back(m - 1, kv.first);
                for (auto it = row.begin(); it != row.end(); ++it) if (f.abs(it->second) == kv.first) {
                    if (f.sign(it->second) < 0) out << '-';
                    out << tev << it->first; row.erase(it); break;
                }
                for (size_t k = 1; k < kv.second; ++k)
                    for (auto it = row.begin(); it != row.end(); ++it) if (f.abs(it->second) == kv.first) {
                        ++nbadd; out << (f.sign(it->second) < 0 ? '-' : '+') << tev << it->first; row.erase(it); break;
                    }
                out << ";\n";
            }
        }
    }
    static void set_transpose(Mat &dst, const Mat &src) { dst = transpose(src); }
    bool triangle(Mat &A, Mat &T, size_t j) {                                                            // :427-507
        if (T.rows[j].empty()) return false;
        bool found = false, over;
        do {
            over = true;
            for (size_t it = 0; it < T.rows[j].size(); ++it) {
                if (absOne(f, T.rows[j][it].second)) continue;
                for (size_t nx = 0; nx < T.rows[j].size(); ++nx) {
                    if (nx == it || absOne(f, T.rows[j][nx].second)) continue;
                    const auto iter = T.rows[j][it], next = T.rows[j][nx];
                    const size_t i = next.first;
                    const E quot = f.div(next.second, iter.second);
                    for (size_t th = 0; th < A.rows[i].size(); ++th) {
                        const auto &third = A.rows[i][th];
                        if (third.first == j || absOne(f, third.second)) continue;
                        if (!absOne(f, f.div(quot, third.second))) continue;
                        size_t m = T.rowdim(); found = true; over = false;
                        out << tev << m << ":=";
                        E ais = f.abs(iter.second);
                        if (f.sign(iter.second) < 0 || f.isMOne(iter.second)) out << '-';
                        f.print_mul(out, tev, j, ais, nbmul); out << ";\n";
                        multiples.push_back(Mult{m, j, iter.second});
                        T.rows.emplace_back();
                        T.rows[m].emplace_back(iter.first, f.one());
                        T.rows[m].emplace_back(next.first, quot);
                        auto &cj = T.rows[j];
                        cj.erase(std::remove_if(cj.begin(), cj.end(), [&](const std::pair<size_t, E> &q) {
                                     return (q.first == iter.first && q.second == iter.second) || (q.first == next.first && q.second == next.second); }), cj.end());
                        set_transpose(A, T);
                        factor_out_rows(A, i);
                        set_transpose(T, A);
                        break;
                    }
                    if (found) break;
                }
                if (found) break;
            }
        } while (!over);
        return found;
    }
    void program_gen() {                                                                                 // :513-611
        Mat T = transpose(M);
        { size_t nc = M.ncols; for (size_t j = 0; j < nc; ++j) factor_out_columns(T, j); }
        set_transpose(M, T);
        for (size_t i = 0; i < M.rowdim(); ++i) factor_out_rows(M, i);
        set_transpose(T, M);
        for (size_t j = 0; j < M.ncols; ++j) triangle(M, T, j);
        for (size_t i = 0; i < M.rowdim(); ++i) {
            const auto &row = M.rows[i];
            if (row.empty()) { out << ouv << i << ":=0;\n"; continue; }
            out << ouv << i << ":=";
            for (size_t k = 0; k < row.size(); ++k) {
                const auto &e = row[k]; E ais = f.abs(e.second); size_t rindex = 0;
                if (k > 0) ++nbadd;
                if (find_mult(e.first, ais, rindex)) {
                    if (k == 0) { if (!(ais == e.second)) out << '-'; } else out << (ais == e.second ? '+' : '-');
                    out << rav << rindex;
                } else {
                    bool ng = f.sign(e.second) < 0 || f.isMOne(e.second);
                    if (k == 0) { if (ng) out << '-'; } else out << (ng ? '-' : '+');
                    f.print_mul(out, tev, e.first, ais, nbmul);
                }
            }
            out << ";\n";
        }
    }
public:
    Replay(const F &ff, const Mat &A, uint64_t seed, std::ostream &os, char o = 'o', char t = 't', char r = 'r')
        : f(ff), M(A), own_rng(seed), rng(own_rng), out(os), ouv(o), tev(t), rav(r) {}
    // the candidate's generator is shared with other Optimizer() calls of the same candidate (LU method)
    Replay(const F &ff, const Mat &A, CandRng &shared, std::ostream &os, char o, char t, char r)
        : f(ff), M(A), own_rng(0), rng(shared), out(os), ouv(o), tev(t), rav(r) {}
    // Optimizer(), include/plinopt_optimize.inl:616-631
    // RecSub's own multiplication count of this schedule (reference include/plinopt_optimize.inl:950-951): naiveOps minus
    // the savings of the steps = the multipliers emitted so far plus every non +-1 entry left, before ProgramGen factors them
    size_t recsub_muls = 0;
    std::pair<size_t, size_t> optimizer() {
        while (one_sub()) {}
        recsub_muls = nbmul; for (auto &r : M.rows) for (auto &e : r) if (!absOne(f, e.second)) ++recsub_muls;
        program_gen(); return {nbadd, nbmul};
    }
    // the fallback of OptMethods :1473-1485: ProgramGen on the untouched matrix
    std::pair<size_t, size_t> direct() { program_gen(); return {nbadd, nbmul}; }
};

// input2Temps with usage check, plinopt_library.inl:319-331
template <class E> void input2temps(std::ostream &os, const SparseMat<E> &M, char inv, char tev) {
    std::vector<char> used(M.coldim(), 0);
    for (auto &r : M.rows) for (auto &e : r) used[e.first] = 1;
    for (size_t j = 0; j < M.coldim(); ++j) if (used[j]) os << tev << j << ":=" << inv << j << ";\n";
}

// Sparse LU with row and column pivoting, M = Qm . L . U . Pm, as LUOptimiser needs it
// (include/plinopt_optimize.inl:1032-1044 calls LinBox GaussDomain::QLUPin, whose pivot choices are not
// specified anywhere in the reference tree).  Pivot rule of this build: the unused row of smallest index that
// still has an entry, and in it the entry of smallest column index.  Conventions of the emitted program
// (:1064-1076): t_k := i_{P[k]};  v := U.t;  x := L.v;  o_i := x_{Q[i]}.
template <class F> struct LUFactors {
    SparseMat<typename F::Elt> U, L;
    std::vector<size_t> P, Q;
    size_t rank = 0;
};
template <class F> LUFactors<F> sparse_lu(const F &f, const SparseMat<typename F::Elt> &M) {
    using E = typename F::Elt;
    const size_t m = M.rowdim(), n = M.coldim();
    std::vector<std::map<size_t, E>> A(m);
    for (size_t i = 0; i < m; ++i) for (auto &e : M.rows[i]) A[i][e.first] = e.second;
    std::vector<char> usedr(m, 0), usedc(n, 0);
    std::vector<size_t> prow, pcol;
    std::vector<std::vector<std::pair<size_t, E>>> mult(m);
    for (;;) {
        size_t r = m;
        for (size_t i = 0; i < m; ++i) if (!usedr[i] && !A[i].empty()) { r = i; break; }
        if (r == m) break;
        const size_t c = A[r].begin()->first, k = prow.size();
        const E piv = A[r].begin()->second;
        usedr[r] = 1; usedc[c] = 1; prow.push_back(r); pcol.push_back(c);
        for (size_t i = 0; i < m; ++i) {
            if (usedr[i]) continue;
            auto it = A[i].find(c);
            if (it == A[i].end()) continue;
            const E l = f.div(it->second, piv);
            for (auto &e : A[r]) {
                E v = f.add(A[i].count(e.first) ? A[i][e.first] : f.zero(), f.neg(f.mul(l, e.second)));
                if (f.isZero(v)) A[i].erase(e.first); else A[i][e.first] = v;
            }
            mult[i].emplace_back(k, l);
        }
    }
    LUFactors<F> R; R.rank = prow.size();
    R.P = pcol; for (size_t j = 0; j < n; ++j) if (!usedc[j]) R.P.push_back(j);
    std::vector<size_t> invP(n); for (size_t k = 0; k < n; ++k) invP[R.P[k]] = k;
    std::vector<size_t> pos(m); size_t nx = R.rank;
    for (size_t k = 0; k < R.rank; ++k) pos[prow[k]] = k;
    for (size_t i = 0; i < m; ++i) if (!usedr[i]) pos[i] = nx++;
    R.Q = pos;
    R.U = SparseMat<E>(m, n); R.L = SparseMat<E>(m, m);
    for (size_t k = 0; k < R.rank; ++k) {
        auto &row = R.U.rows[k];
        for (auto &e : A[prow[k]]) row.emplace_back(invP[e.first], e.second);
        std::sort(row.begin(), row.end(), [](const std::pair<size_t, E> &x, const std::pair<size_t, E> &y) { return x.first < y.first; });
    }
    for (size_t i = 0; i < m; ++i) {
        auto &row = R.L.rows[pos[i]];
        for (auto &e : mult[i]) row.emplace_back(e.first, e.second);
        if (usedr[i]) row.emplace_back(pos[i], f.one());
    }
    return R;
}

// --------------------------------------------------------------------- -A: M = Alt . CoB
// backSolver / Factorizer (reference include/plinopt_sparsify.inl:756-867, :924-984) for the inner dimension
// backSolver (reference include/plinopt_sparsify.inl:756-867) for an inner dimension k, n = coldim(M) <= k <= rowdim(M)
// (ABOptimiser tries every k in [n, m), plinopt_optimize.inl:1437-1439): a random order of the rows (the build's
// per-candidate stream; the reference uses LinBox's Permutation::random), the first n independent rows in that order are
// swapped to the front as :784-799 does, the next k-n rows of the resulting order join them (:802-804): these k rows are
// CoB (k x n) and unit rows of Alt (m x k, :842); every other row is solved for, x . CoB = row (:845-851).  For k > n the
// system is underdetermined and the reference takes whatever particular solution LinBox's QLUP solve returns (its pivoting
// is not in the tree); here the solution is the one supported on the n independent rows (the added rows get coefficient 0).
template <class F> struct ABFactors {
    SparseMat<typename F::Elt> Alt, CoB;
    std::array<size_t, 3> score{0, 0, 0};     // nnz(Alt), non +-1 entries of Alt, nnz(CoB): tricOpCount :911-920
    bool identity = false;
};
template <class F> bool ab_backsolve(const F &f, const SparseMat<typename F::Elt> &M, uint64_t seed, ABFactors<F> &out, size_t innerdim = 0) {
    using E = typename F::Elt;
    const size_t m = M.rowdim(), n = M.coldim(), kdim = innerdim ? innerdim : n;
    if (kdim < n || kdim > m) return false;
    std::vector<size_t> ord(m);
    for (size_t i = 0; i < m; ++i) ord[i] = i;
    CandRng rng(seed);
    for (size_t i = m; i > 1; --i) std::swap(ord[i - 1], ord[rng.next() % (uint32_t)i]);
    // greedy row basis in that order; ech keeps the echelon rows together with the combination of basis rows they are
    std::vector<std::vector<E>> ech, comb; std::vector<size_t> piv;
    auto dense_row = [&](size_t r) { std::vector<E> v(n, f.zero()); for (auto &e : M.rows[r]) v[e.first] = e.second; return v; };
    size_t nb = 0;
    for (size_t i = 0; i < n; ++i) {                              // :784-799: position i takes the first later row that raises the rank
        bool got = false;
        for (size_t j = i; j < m && !got; ++j) {
            std::vector<E> v = dense_row(ord[j]), c(n, f.zero());
            for (size_t k = 0; k < ech.size(); ++k) {
                const E x = v[piv[k]];
                if (f.isZero(x)) continue;
                for (size_t q = 0; q < n; ++q) v[q] = f.add(v[q], f.neg(f.mul(x, ech[k][q])));
                for (size_t q = 0; q < n; ++q) c[q] = f.add(c[q], f.neg(f.mul(x, comb[k][q])));
            }
            size_t pc = n;
            for (size_t q = 0; q < n; ++q) if (!f.isZero(v[q])) { pc = q; break; }
            if (pc == n) continue;                                // dependent on the rows chosen so far
            const E iv = f.inv(v[pc]);
            c[i] = f.add(c[i], f.one());                          // this row itself is basis row number i
            for (size_t q = 0; q < n; ++q) { v[q] = f.mul(v[q], iv); c[q] = f.mul(c[q], iv); }
            ech.push_back(v); comb.push_back(c); piv.push_back(pc);
            std::swap(ord[i], ord[j]); got = true; ++nb;
        }
        if (!got) break;
    }
    if (nb < n) return false;                                     // rank < n: no change of basis of full column rank
    out.CoB = SparseMat<E>(kdim, n); out.Alt = SparseMat<E>(m, kdim);
    for (size_t k = 0; k < kdim; ++k) { out.CoB.rows[k] = M.rows[ord[k]]; out.Alt.rows[ord[k]].emplace_back(k, f.one()); }   // :802-804, :842
    for (size_t t = kdim; t < m; ++t) {                           // x . CoB = row, supported on the n independent rows
        const size_t i = ord[t];
        std::vector<E> v = dense_row(i), x(n, f.zero());
        for (size_t k = 0; k < n; ++k) {
            const E y = v[piv[k]];
            if (f.isZero(y)) continue;
            for (size_t q = 0; q < n; ++q) v[q] = f.add(v[q], f.neg(f.mul(y, ech[k][q])));
            for (size_t q = 0; q < n; ++q) x[q] = f.add(x[q], f.mul(y, comb[k][q]));
        }
        for (size_t q = 0; q < n; ++q) if (!f.isZero(v[q])) return false;      // cannot happen at full column rank
        for (size_t q = 0; q < n; ++q) if (!f.isZero(x[q])) out.Alt.rows[i].emplace_back(q, x[q]);
    }
    size_t nz = 0, nu = 0, nc = 0;
    for (auto &r : out.Alt.rows) for (auto &e : r) { ++nz; if (!absOne(f, e.second)) ++nu; }
    for (auto &r : out.CoB.rows) nc += r.size();
    out.score = {nz, nu, nc};
    return true;
}
// Factorizer :924-984: best of `loops` back-solves (seeds seed0 ...), starting from (Alt, CoB) = (M, identity)
template <class F> ABFactors<F> ab_factorize(const F &f, const SparseMat<typename F::Elt> &M, size_t loops, uint64_t seed0, size_t innerdim = 0) {
    using E = typename F::Elt;
    const size_t m = M.rowdim(), n = M.coldim(), kdim = innerdim ? innerdim : n;
    ABFactors<F> best;
    if (m == n) {                                                // :945-951 identity factorization
        best.CoB = M; best.Alt = SparseMat<E>(m, n); best.identity = true;
        for (size_t i = 0; i < m; ++i) best.Alt.rows[i].emplace_back(i, f.one());
        return best;
    }
    best.Alt = SparseMat<E>(m, kdim); for (size_t i = 0; i < m; ++i) best.Alt.rows[i] = M.rows[i];     // :953-955: (M | 0) . (I ; 0)
    best.CoB = SparseMat<E>(kdim, n);
    for (size_t i = 0; i < n; ++i) best.CoB.rows[i].emplace_back(i, f.one());
    size_t nz = 0, nu = 0;
    for (auto &r : M.rows) for (auto &e : r) { ++nz; if (!absOne(f, e.second)) ++nu; }
    best.score = {nz, nu, n};
    if (m < n) return best;
    size_t bi = loops;                                           // ties go to the smaller seed, whatever the thread schedule
#pragma omp parallel for schedule(dynamic, 16)
    for (long long i = 0; i < (long long)loops; ++i) {
        ABFactors<F> c;
        if (!ab_backsolve(f, M, seed0 + (uint64_t)i, c, kdim)) continue;
#pragma omp critical
        if (c.score < best.score || (c.score == best.score && bi != loops && (size_t)i < bi)) { best = c; bi = (size_t)i; }
    }
    return best;
}

// --------------------------------------------------------------------- -K: kernel (nullspace) decomposition
// nullspacedecomp (reference include/plinopt_optimize.inl:689-884) computes a row basis of M through LinBox's
// InPlaceLinearPivoting and a nullspace basis, none of whose choices are specified in the tree.  This build's rule,
// from one per-decomposition stream: a random order of the rows (Fisher-Yates, as :705-713 shuffles), the greedy row
// basis in that order, every other row d as its combination x_d of the basis rows; NotIndep = next() mod nullity of
// the dependent rows stay in the directly computed part (:789-797, the LAST ones in the order), the others are emptied in
// `Free` and computed by `Dep` (rows = those dependent rows, columns = rows of M, entries on basis rows only):
//     o := Free . i ;   x := Dep . o ;   o_{dep[j]} := x_j                    (:836-872)
template <class F> struct KernelDecomp {
    SparseMat<typename F::Elt> Free, Dep;
    std::vector<size_t> dep;                  // row of M computed by row j of Dep
    size_t rank = 0, notindep = 0;
};
template <class F> bool kernel_decomp_order(const F &f, const SparseMat<typename F::Elt> &M, const std::vector<size_t> &ord, CandRng &rng, KernelDecomp<F> &out);
template <class F> bool kernel_decomp(const F &f, const SparseMat<typename F::Elt> &M, uint64_t seed, KernelDecomp<F> &out) {
    const size_t m = M.rowdim();
    std::vector<size_t> ord(m);
    for (size_t i = 0; i < m; ++i) ord[i] = i;
    CandRng rng(seed);
    for (size_t i = m; i > 1; --i) std::swap(ord[i - 1], ord[rng.next() % (uint32_t)i]);
    return kernel_decomp_order(f, M, ord, rng, out);
}
// the same with a prescribed order of the rows (-N, AllKernelOpt :1357-1418 walks all of them)
template <class F> bool kernel_decomp_order(const F &f, const SparseMat<typename F::Elt> &M, const std::vector<size_t> &ord, CandRng &rng, KernelDecomp<F> &out) {
    using E = typename F::Elt;
    const size_t m = M.rowdim(), n = M.coldim();
    std::vector<std::vector<E>> ech, comb; std::vector<size_t> piv, basis, deps;
    std::vector<std::vector<E>> depx;
    for (size_t t = 0; t < m; ++t) {
        std::vector<E> v(n, f.zero()), c(m, f.zero());        // c: combination over basis rows, indexed by basis position
        for (auto &e : M.rows[ord[t]]) v[e.first] = e.second;
        for (size_t k = 0; k < ech.size(); ++k) {
            const E x = v[piv[k]];
            if (f.isZero(x)) continue;
            for (size_t j = 0; j < n; ++j) v[j] = f.add(v[j], f.neg(f.mul(x, ech[k][j])));
            for (size_t j = 0; j < ech.size(); ++j) c[j] = f.add(c[j], f.mul(x, comb[k][j]));
        }
        size_t pc = n;
        for (size_t j = 0; j < n; ++j) if (!f.isZero(v[j])) { pc = j; break; }
        if (pc == n) { deps.push_back(ord[t]); depx.push_back(c); continue; }      // row = sum_j c[j] * basis row j
        // new basis row number |basis|: echelon row = (row - sum c_j basis_j) / pivot
        const E iv = f.inv(v[pc]);
        std::vector<E> cc(m, f.zero());
        for (size_t j = 0; j < ech.size(); ++j) cc[j] = f.neg(f.mul(c[j], iv));
        cc[basis.size()] = iv;
        for (size_t j = 0; j < n; ++j) v[j] = f.mul(v[j], iv);
        ech.push_back(v); comb.push_back(cc); piv.push_back(pc); basis.push_back(ord[t]);
    }
    if (deps.empty()) return false;                             // zero dimensional kernel (:1304-1309)
    out.rank = basis.size();
    out.notindep = rng.next() % (uint32_t)deps.size();          // :792-795
    const size_t kept = deps.size() - out.notindep;
    out.Free = M; out.Dep = SparseMat<E>(kept, m); out.dep.assign(deps.begin(), deps.begin() + (long)kept);
    for (size_t j = 0; j < kept; ++j) {
        out.Free.rows[deps[j]].clear();
        std::vector<std::pair<size_t, E>> row;
        for (size_t b = 0; b < basis.size(); ++b) if (!f.isZero(depx[j][b])) row.emplace_back(basis[b], depx[j][b]);
        std::sort(row.begin(), row.end(), [](const std::pair<size_t, E> &x, const std::pair<size_t, E> &y) { return x.first < y.first; });
        out.Dep.rows[j] = row;
    }
    return true;
}

// cmpOpCount, include/plinopt_optimize.h:53-64 (mode 0 default, 1 OPTIMIZE_ADDITIONS, 2 OPTIMIZE_SUMS)
inline bool cmp_op_count(std::pair<size_t, size_t> a, std::pair<size_t, size_t> b, int mode = 0) {
    if (mode == 1) return a.first < b.first || (a.first == b.first && a.second < b.second);
    if (mode == 2) return a.first + a.second < b.first + b.second;
    size_t as = a.first + a.second, bs = b.first + b.second;
    return as < bs || (as == bs && a.first < b.first);
}


// ---------------------------------------------------------------------------------------------------------------
// Seed shards over several GPUs of one node from a command-line tool: one forked child per device (`--gpu N`, N >= 2).
// The candidates of a restart loop are independent (reference: iterations of `#pragma omp parallel for`,
// include/plinopt_optimize.inl:1204-1205, plinopt_inplace.inl:837-838), so rank r takes the r-th contiguous block of
// the seed range and the parent keeps the minimum under the tools' total order (cost, seed): the winner is the one a
// single device finds.  Children are forked BEFORE this process touches the HIP runtime (a runtime does not survive
// fork) and each opens its own device; results come back over a pipe as plain structs.
// PLO_GPU_DEVICES="0,0,1": device ordinal per shard (default: shard r on device r) -- lets a one-GPU box run N shards.
struct ShardOut { int32_t ok; int32_t rc; uint32_t a, b, c; uint64_t seed; uint64_t variant; uint64_t candidates; double kernel_ms; char msg[192]; };   // rc: the library's return code of a failed shard (0 otherwise)
inline void shard_block(uint64_t seed0, uint64_t n, int rank, int world, uint64_t &s, uint64_t &cnt) {
    const uint64_t q = n / (uint64_t)world, r = n % (uint64_t)world;
    s = seed0 + (uint64_t)rank * q + std::min<uint64_t>((uint64_t)rank, r); cnt = q + ((uint64_t)rank < r ? 1 : 0);
}
inline int shard_device(int rank) {
    if (const char *e = getenv("PLO_GPU_DEVICES")) {
        std::string t(e); size_t pos = 0; int k = 0;
        while (pos <= t.size()) { size_t c = t.find(',', pos); if (c == std::string::npos) c = t.size(); if (k == rank) return atoi(t.substr(pos, c - pos).c_str()); pos = c + 1; ++k; }
    }
    return rank;
}
// fn(rank, device, first seed, count) -> ShardOut, run in a child process per shard; returns false when a child failed
template <class Fn> bool forked_shards(int world, uint64_t seed0, uint64_t n, Fn fn, std::vector<ShardOut> &out) {
    out.assign((size_t)world, ShardOut{});
    std::vector<int> fds((size_t)world, -1); std::vector<pid_t> pids((size_t)world, -1);
    // a failed pipe() or fork(): close what was opened, reap the children already started (they finish their shard: a child is never
    // left running on its GPU as an orphan) and say why
    auto abandon = [&](int upto, const char *what) {
        const int err = errno;
        for (int k = 0; k < upto; ++k) { if (fds[(size_t)k] >= 0) close(fds[(size_t)k]); if (pids[(size_t)k] > 0) { kill(pids[(size_t)k], SIGTERM); int st = 0; waitpid(pids[(size_t)k], &st, 0); } }
        for (auto &o : out) { o.ok = 0; snprintf(o.msg, sizeof o.msg, "%s: %s", what, strerror(err)); }
        return false;
    };
    for (int r = 0; r < world; ++r) {
        int pfd[2]; if (pipe(pfd) != 0) return abandon(r, "pipe");
        std::cout.flush(); std::clog.flush();
        const pid_t pid = fork();
        if (pid < 0) { close(pfd[0]); close(pfd[1]); return abandon(r, "fork"); }
        if (pid == 0) {
            close(pfd[0]);
            uint64_t s, cnt; shard_block(seed0, n, r, world, s, cnt);
            ShardOut o{};
            try { o = fn(r, shard_device(r), s, cnt); } catch (const std::exception &e) { o.ok = 0; snprintf(o.msg, sizeof o.msg, "%s", e.what()); }
            ssize_t w = write(pfd[1], &o, sizeof o); (void)w;
            close(pfd[1]);
            _exit(0);
        }
        close(pfd[1]); fds[(size_t)r] = pfd[0]; pids[(size_t)r] = pid;
    }
    bool good = true;
    for (int r = 0; r < world; ++r) {
        size_t got = 0; char *dst = (char *)&out[(size_t)r];
        while (got < sizeof(ShardOut)) { ssize_t k = read(fds[(size_t)r], dst + got, sizeof(ShardOut) - got); if (k <= 0) break; got += (size_t)k; }
        close(fds[(size_t)r]);
        int status = 0; waitpid(pids[(size_t)r], &status, 0);
        if (got != sizeof(ShardOut) || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { out[(size_t)r].ok = 0; if (!out[(size_t)r].msg[0]) snprintf(out[(size_t)r].msg, sizeof out[(size_t)r].msg, "shard process died"); }
        if (!out[(size_t)r].ok) good = false;
    }
    return good;
}
} // namespace plo
#endif
