// ===========================================================================
// plo_compact.hpp -- the rewriting engine of bin/compacter (see compacter.cpp for the
// contract; reference src/compacter.cpp:27-68, include/plinopt_programs.inl:1157-1455),
// also used by `bin/optimizer -F` to clean the program of the kernel method with
// identity goals (reference include/plinopt_optimize.inl:637-685).
// ===========================================================================
#ifndef PLO_COMPACT_HPP
#define PLO_COMPACT_HPP
#include "plo_host.hpp"
#include <functional>

namespace plo { namespace compact {
using IsOutput = std::function<bool(const std::string &)>;
using Toks = std::vector<std::string>;
struct Line { std::string lhs; Toks rhs; };          // rhs without the final ';'

inline bool is_var(const std::string &t) { return !t.empty() && (isalpha((unsigned char)t[0]) || t[0] == '_'); }
inline IsOutput letter_outputs(char ouv) { return [ouv](const std::string &v) { return v[0] == ouv && v.size() > 1 && isdigit((unsigned char)v[1]); }; }

inline std::vector<Line> parse(std::istream &in) {
    std::vector<Line> P; std::string line;
    while (std::getline(in, line)) {
        auto h = line.find('#'); if (h != std::string::npos) line.resize(h);
        // several statements may share a line; each ends at ';'
        size_t start = 0;
        while (start < line.size()) {
            size_t semi = line.find(';', start);
            std::string st = line.substr(start, semi == std::string::npos ? std::string::npos : semi - start);
            start = semi == std::string::npos ? line.size() : semi + 1;
            if (st.find(":=") == std::string::npos) continue;
            Toks tk = SlpEval<QField>::tokenize(st);
            if (tk.size() < 3 || tk[1] != ":=" || !is_var(tk[0])) throw std::runtime_error("compacter: bad statement: " + st);
            P.push_back(Line{tk[0], Toks(tk.begin() + 2, tk.end())});
        }
    }
    return P;
}
inline size_t prog_size(const std::vector<Line> &P) { size_t s = 0; for (auto &l : P) s += l.rhs.size() + 2; return s; }   // elements: lhs, :=, rhs words (progSize)

inline bool has_toplevel_sum(const Toks &e) {                // a + or - that is not a leading sign and not inside parentheses
    int depth = 0;
    for (size_t k = 0; k < e.size(); ++k) {
        if (e[k] == "(") ++depth; else if (e[k] == ")") --depth;
        else if (depth == 0 && k > 0 && (e[k] == "+" || e[k] == "-") && e[k - 1] != "*" && e[k - 1] != "/" && e[k - 1] != "(") return true;
    }
    return false;
}
// e with all its top-level signs flipped, as a sequence that starts with an explicit sign
inline Toks signed_terms(const Toks &e, bool negate) {
    Toks o; int depth = 0;
    for (size_t k = 0; k < e.size(); ++k) {
        const std::string &t = e[k];
        const bool top_sign = depth == 0 && (t == "+" || t == "-") && (k == 0 || (e[k - 1] != "*" && e[k - 1] != "/" && e[k - 1] != "("));
        if (k == 0 && !top_sign) o.push_back(negate ? "-" : "+");
        if (t == "(") ++depth; else if (t == ")") --depth;
        if (top_sign) o.push_back((t == "-") != negate ? "-" : "+"); else o.push_back(t);
    }
    return o;
}

struct Info { int defs = 0, uses = 0; std::vector<size_t> lines; };       // lines: where it is used (one entry per occurrence)
inline std::map<std::string, Info> census(const std::vector<Line> &P) {
    std::map<std::string, Info> I;
    for (size_t k = 0; k < P.size(); ++k) { ++I[P[k].lhs].defs; for (auto &t : P[k].rhs) if (is_var(t)) { auto &e = I[t]; ++e.uses; e.lines.push_back(k); } }
    return I;
}
// every variable of e keeps its value from line `from` to line `to` (it is an input, or assigned once and before `from`)
inline bool stable(const std::vector<Line> &P, const std::map<std::string, Info> &I, const Toks &e, size_t from, size_t to) {
    for (auto &t : e) if (is_var(t)) {
        auto it = I.find(t);
        if (it == I.end() || it->second.defs == 0) continue;
        if (it->second.defs == 1) { bool later = false; for (size_t k = from + 1; k <= to && k < P.size(); ++k) if (P[k].lhs == t) { later = true; break; } if (later) return false; continue; }
        for (size_t k = from + 1; k <= to && k < P.size(); ++k) if (P[k].lhs == t) return false;
    }
    return true;
}

// one round over the whole program; returns true when something changed
inline bool trim(std::vector<Line> &P, bool simplSingle, const IsOutput &is_output) {
    bool changed = false;
    auto I = census(P);
    std::vector<char> gone(P.size(), 0);
    // no-ops (x:=x) and dead temporaries
    for (size_t k = 0; k < P.size(); ++k) {
        const Line &l = P[k];
        const bool noop = l.rhs.size() == 1 && l.rhs[0] == l.lhs;
        const bool dead = !is_output(l.lhs) && I[l.lhs].uses == 0;
        if (noop || dead) { gone[k] = 1; changed = true; }
    }
    if (changed) { std::vector<Line> Q; for (size_t k = 0; k < P.size(); ++k) if (!gone[k]) Q.push_back(std::move(P[k])); P.swap(Q); return true; }
    // copies and single uses.  A substitution moves the occurrences of E's variables from the definition to the use
    // (their counts and, for a single use, the census stay valid), so one round takes every candidate in program order.
    for (size_t d = 0; d < P.size(); ++d) {
        if (gone[d]) continue;
        const std::string x = P[d].lhs;
        Info &ix = I[x];
        if (is_output(x) || ix.defs != 1 || ix.uses == 0) continue;
        bool selfref = false; for (auto &t : P[d].rhs) if (t == x) selfref = true;
        if (selfref) continue;
        const Toks E = P[d].rhs;
        const bool copy = E.size() == 1 && (is_var(E[0]) || isdigit((unsigned char)E[0][0]));
        if (!copy && !(simplSingle && ix.uses == 1)) continue;
        size_t last = d; bool before = false;
        for (size_t k : ix.lines) { if (k <= d) before = true; last = std::max(last, k); }
        if (before || last == d || !stable(P, I, E, d, last)) continue;
        const bool sum = has_toplevel_sum(E), lead = !E.empty() && (E[0] == "-" || E[0] == "+");
        std::vector<size_t> where(ix.lines); std::sort(where.begin(), where.end()); where.erase(std::unique(where.begin(), where.end()), where.end());
        for (size_t k : where) {
            Toks &R = P[k].rhs; Toks N;
            for (size_t z = 0; z < R.size(); ++z) {
                if (R[z] != x) { N.push_back(R[z]); continue; }
                const std::string prev = z ? R[z - 1] : ":=", next = z + 1 < R.size() ? R[z + 1] : ";";
                const bool mult_ctx = prev == "*" || prev == "/" || next == "*" || next == "/";
                if (copy) { N.push_back(E[0]); continue; }
                if (mult_ctx) {
                    // a product chain may continue a product on its left (a*x*3 with x = b*2) but not a quotient, and nothing signed or summed may
                    const bool chain_ok = !sum && !lead && prev != "/" && !(prev == "*" && std::find(E.begin(), E.end(), "/") != E.end());
                    if (chain_ok) N.insert(N.end(), E.begin(), E.end());
                    else { N.push_back("("); N.insert(N.end(), E.begin(), E.end()); N.push_back(")"); }
                    continue;
                }
                // additive context: fold the signs
                if (prev == "+" || prev == "-") {
                    Toks S = signed_terms(E, prev == "-");
                    N.pop_back();                                  // the use's own sign is replaced by the first sign of S
                    if (S[0] == "+" && (N.empty() || N.back() == "(")) S.erase(S.begin());   // no unary plus after := or (
                    N.insert(N.end(), S.begin(), S.end());
                } else {                                           // after := or (
                    N.insert(N.end(), E.begin(), E.end());
                }
            }
            R.swap(N);
        }
        if (copy && is_var(E[0])) { Info &iy = I[E[0]]; iy.uses += ix.uses - 1; for (size_t k : ix.lines) iy.lines.push_back(k); }
        ix.uses = 0; ix.lines.clear();
        gone[d] = 1; changed = true;
    }
    if (changed) { std::vector<Line> Q; for (size_t k = 0; k < P.size(); ++k) if (!gone[k]) Q.push_back(std::move(P[k])); P.swap(Q); return true; }
    // leading minus behind a positive term: x:=-a+b -> x:=b-a
    for (auto &l : P) {
        Toks &R = l.rhs;
        if (R.size() < 4 || R[0] != "-") continue;
        int depth = 0; size_t plus = 0;
        for (size_t k = 1; k < R.size(); ++k) {
            if (R[k] == "(") ++depth; else if (R[k] == ")") --depth;
            else if (depth == 0 && R[k] == "+" && R[k - 1] != "*" && R[k - 1] != "/" && R[k - 1] != "(") { plus = k; break; }
        }
        if (!plus) continue;
        // the positive term runs from plus+1 to the next top-level sign
        size_t end = R.size(); depth = 0;
        for (size_t k = plus + 1; k < R.size(); ++k) {
            if (R[k] == "(") ++depth; else if (R[k] == ")") --depth;
            else if (depth == 0 && (R[k] == "+" || R[k] == "-") && R[k - 1] != "*" && R[k - 1] != "/" && R[k - 1] != "(") { end = k; break; }
        }
        Toks N(R.begin() + (long)plus + 1, R.begin() + (long)end);
        N.insert(N.end(), R.begin(), R.begin() + (long)plus);
        N.insert(N.end(), R.begin() + (long)end, R.end());
        R.swap(N);
        changed = true;
    }
    return changed;
}

inline void print(std::ostream &os, const std::vector<Line> &P) {
    for (auto &l : P) { os << l.lhs << ":="; for (auto &t : l.rhs) os << t; os << ";\n"; }
}

// all rounds
inline void compact_program(std::vector<Line> &P, bool simplSingle, const IsOutput &is_output, size_t maxrounds = 0) {
    size_t rounds = 0;
    while (trim(P, simplSingle, is_output)) { if (maxrounds && ++rounds >= 2 * maxrounds) break; }
}
} } // namespace plo::compact
#endif
