// ===========================================================================
// plo_trim.hpp -- the rewriting engine of bin/compacter.
//
// Reference: `Compacter` src/compacter.cpp:27-68, `programParser`
// include/plinopt_programs.inl:618-686, `variablesTrimer` :1157-1455 and the sign /
// parenthesis helpers :693-1142.  Every reference pipeline pipes the optimizer's program
// through `compacter -s` before checking it (bin/FDT.sh:58) and bin/GDT.sh:41-65 pins the
// operation count of `compacter f` for every stored data/*.slp, so this tool has to print
// the reference's TEXT, not just an equivalent program: the passes below do, in the
// reference's order and with its tie rules, what variablesTrimer does
//   [1] outputs renamed to a free letter, `o_k := z_k;` appended            (:1169-1187)
//   [2] copies `x := a;` replaced by `a` until x is assigned again          (:1189-1206)
//   [3] the appended copies folded back into the last assignment            (:1212-1232)
//   [-] leading minus rotated behind a positive group, temporaries negated
//       when that removes the leading minus of an output                    (:1240-1244)
//   [4] a temporary used once is written in place of its use, with the sign
//       and parenthesis rules of :1308-1358; constant factors combined      (:1361-1405)
//   [5] minus signs moved across parentheses, temporaries negated when that
//       lowers the number of lines starting with `-`                        (:1423-1449)
// but on interned words (a line is a vector of small integers, the eight operator
// words have fixed numbers) with splice-style rewrites instead of the reference's
// string vectors.  Held line by line to the literal restatement
// oracle/plo_compact_oracle.py (tests/test_compacter.py).
//
// Not pinned (Givaro is not in the tree): what `Givaro::Rational(const char*)` makes of a
// word that is not a number when two `*`/`/` stand a word apart (:1367-1371); this engine
// leaves such a chain as it is.  Constant products beyond 127 bits raise an error.
// A malformed line (the reference would read out of range) raises an error too.
// ===========================================================================
#ifndef PLO_TRIM_HPP
#define PLO_TRIM_HPP
#include <algorithm>
#include <cstdint>
#include <istream>
#include <map>
#include <ostream>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace plo { namespace trim {

enum : int { W_ASSIGN = 0, W_SEMI, W_PLUS, W_MINUS, W_MUL, W_DIV, W_LP, W_RP, W_EMPTY, W_NAMED };
using Line = std::vector<int>;
using Prog = std::vector<Line>;

struct Words {                                         // the word table: text <-> number
    std::vector<std::string> text;
    std::vector<char> plain;                           // no operator character inside (isVariable, :99-100)
    std::unordered_map<std::string, int> number;
    Words() { for (const char *w : {":=", ";", "+", "-", "*", "/", "(", ")", ""}) get(w); }
    int get(const std::string &s) {
        auto it = number.find(s);
        if (it != number.end()) return it->second;
        const int k = (int)text.size();
        text.push_back(s); plain.push_back(s.find_first_of("+-*/;:=()") == std::string::npos);
        number.emplace(s, k);
        return k;
    }
    char first(int w) const { return text[w].empty() ? '\0' : text[w][0]; }
};

inline bool addsub(int w) { return w == W_PLUS || w == W_MINUS; }
inline bool muldiv(int w) { return w == W_MUL || w == W_DIV; }
inline bool opener(int w) { return w == W_LP || w == W_ASSIGN; }       // isParAff, :96-97
inline int flipped(int w) { return w == W_PLUS ? W_MINUS : w == W_MINUS ? W_PLUS : w; }
inline int at(const Line &L, size_t k) { if (k >= L.size()) throw std::runtime_error("compacter: malformed line"); return L[k]; }
inline bool copy_line(const Line &L) { return L.size() == 4 && L[0] == L[2]; }        // `x := x ;`, :77-78
inline size_t elements(const Prog &P) { size_t s = 0; for (auto &l : P) s += l.size(); return s - 2 * P.size(); }   // progSize :110-114

// counted operations of a program (lineOperations / progOperations, :116-141): a sign right behind `:=` or `(` is a
// negation, every other sign an addition, every `*` or `/` between words a multiplication
inline std::pair<size_t, size_t> operations(const Prog &P) {
    size_t adds = 0, muls = 0;
    for (auto &L : P) { bool neg = false; for (int w : L) { if (addsub(w) && !neg) ++adds; else if (muldiv(w)) ++muls; neg = opener(w); } }
    return {adds, muls};
}

// ---- reading and printing -----------------------------------------------------------------
inline bool digits_only(const std::string &s) { for (unsigned char c : s) if (!isdigit(c)) return false; return true; }

inline Prog parse(std::istream &in, Words &W) {
    Prog P; std::string s;
    auto delim = [](char c, bool slash) { return c == '(' || c == ')' || c == '+' || c == '-' || c == '*' || c == ';' || (slash && c == '/'); };
    while (std::getline(in, s)) {
        const size_t h = s.find('#'); if (h != std::string::npos) s.resize(h);
        while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
        const size_t a = s.find(":=");
        if (a == std::string::npos) continue;
        Line L{W.get(s.substr(0, a)), W_ASSIGN};
        size_t b = a + 2;                              // start of the word being read
        for (size_t k = b; k < s.size(); ++k) {
            if (!delim(s[k], true)) continue;
            if (k > b) {
                std::string word = s.substr(b, k - b);
                if (s[k] == '/' && digits_only(word)) {          // a rational constant n/d is ONE word (:645-658)
                    size_t e = k + 1; while (e < s.size() && !delim(s[e], false)) ++e;
                    if (e >= s.size()) throw std::runtime_error("compacter: constant without a delimiter behind it: " + s);
                    word = s.substr(b, e - b); k = e;
                }
                L.push_back(W.get(word));
            }
            L.push_back(W.get(std::string(1, s[k])));
            b = k + 1;
        }
        if (b < s.size()) L.push_back(W.get(s.substr(b)));
        P.push_back(std::move(L));
    }
    return P;
}
inline void print(std::ostream &os, const Prog &P, const Words &W) {
    for (auto &L : P) { if (L.empty()) continue; for (int w : L) os << W.text[w]; os << '\n'; }
}

// ---- signs ----------------------------------------------------------------------------------
// `x := -a ... + g ... ;`  ->  `x := g ... -a ... ;` : the first group behind a top-level `+` comes first (:693-723)
inline bool lead_with_plus_group(Line &L) {
    if (at(L, 2) != W_MINUS) return false;
    long depth = 0;
    for (size_t k = 3; k < L.size(); ++k) {
        if (L[k] == W_LP) ++depth; else if (L[k] == W_RP) --depth;
        if (depth == 0 && L[k] == W_PLUS) {
            Line N(L.begin(), L.begin() + 2);
            N.insert(N.end(), L.begin() + k + 1, L.end() - 1);
            N.insert(N.end(), L.begin() + 2, L.begin() + k);
            N.push_back(L.back());
            L.swap(N);
            return true;
        }
    }
    return false;
}
// the same inside a parenthesis that opens at `open`, taking the LAST top-level `+` of the group (:726-753)
inline bool group_lead_with_plus(Line &B, size_t open) {
    if (at(B, open) != W_LP || at(B, open + 1) != W_MINUS) return false;
    long depth = 0; size_t plus = 0, close = 0;
    for (size_t k = open + 2; k < B.size(); ++k) {
        if (B[k] == W_LP) ++depth; else if (B[k] == W_RP) --depth;
        if (depth < 0) { close = k; break; }
        if (depth == 0 && B[k] == W_PLUS) plus = k;
    }
    if (!plus) return false;
    if (!close) throw std::runtime_error("compacter: parenthesis not closed");
    Line N(B.begin(), B.begin() + open + 1);
    N.insert(N.end(), B.begin() + plus + 1, B.begin() + close);
    N.insert(N.end(), B.begin() + open + 1, B.begin() + plus);
    N.insert(N.end(), B.begin() + close, B.end());
    B.swap(N);
    return true;
}
// minus the right-hand side: every top-level sign flips, a missing first sign becomes `-`, a `-` right behind the
// opening word disappears (:757-789).  The first two words are the head (`x :=` or `sign (`).
inline Line negated(const Line &L) {
    Line N{at(L, 0), at(L, 1)};
    N.reserve(L.size() + 1);
    if (at(L, 2) != W_MINUS) N.push_back(W_MINUS);
    long depth = 0;
    for (size_t k = 2; k < L.size(); ++k) {
        const int w = L[k];
        if (w == W_LP) ++depth; else if (w == W_RP) --depth;
        if (depth == 0 && w == W_MINUS) { if (!opener(N.back())) N.push_back(W_PLUS); }
        else if (depth == 0 && w == W_PLUS) N.push_back(W_MINUS);
        else N.push_back(w);
    }
    return N;
}
// the sign in front of every occurrence of `var` flips (:794-821)
inline bool negate_uses(Line &L, int var) {
    bool hit = false;
    for (size_t j = 2; j < L.size(); ++j) {
        if (L[j] != var) continue;
        hit = true;
        if (L[j - 1] == W_PLUS) L[j - 1] = W_MINUS;
        else if (L[j - 1] == W_MINUS) {
            if (opener(L[j - 2])) L.erase(L.begin() + j - 1);      // the scan goes on one word further right, as :806-808 does
            else L[j - 1] = W_PLUS;
        } else if (opener(L[j - 1])) { L.insert(L.begin() + j, W_MINUS); ++j; }
    }
    return hit;
}
inline bool drop_plus_after_assign(Line &L) {          // `x := + a` -> `x := a` (:905-911)
    if (L.size() > 2 && L[1] == W_ASSIGN && L[2] == W_PLUS) { L.erase(L.begin() + 2); return true; }
    return false;
}
inline size_t leading_minus_lines(const Prog &P) { size_t c = 0; for (auto &L : P) c += at(L, 2) == W_MINUS; return c; }

struct Engine {
    Words W;
    char inchar = 'i', outchar = 'o';

    // An output line that still starts with `-`: negate one of its temporaries (definitions above, uses below) when the
    // number of leading minus signs does not grow -- or whatever it costs with `force` (:828-883)
    bool fix_output_line(Prog &P, size_t index, bool force) {
        if (W.first(at(P[index], 0)) != outchar) return false;
        lead_with_plus_group(P[index]);
        if (at(P[index], 2) != W_MINUS) return false;
        for (size_t j = 3; j < P[index].size(); ++j) {
            const int var = P[index][j];
            if (!W.plain[var] || W.first(var) == inchar || W.first(var) == outchar) continue;
            Prog N(P);
            long balance = 1;
            for (size_t k = index; k-- > 0;) {
                if (N[k][0] != var) continue;
                const int before = at(N[k], 2) == W_MINUS;
                Line def = negated(N[k]);
                lead_with_plus_group(def);
                const int delta = (at(def, 2) == W_MINUS) - before;
                if (!force && delta > 0) break;
                balance += delta;
                N[k].swap(def);
                for (size_t l = k + 1; l < N.size(); ++l) {
                    const int was = at(N[l], 2) == W_MINUS;
                    negate_uses(N[l], var);
                    lead_with_plus_group(N[l]);
                    balance += (at(N[l], 2) == W_MINUS) - was;
                    if (N[l][0] == var) break;
                }
            }
            if (force || balance <= 0) { P.swap(N); return true; }
        }
        return false;
    }
    void fix_output_lines(Prog &P, bool force) {       // from the last line up (:885-899)
        for (size_t i = P.size(); i-- > 0;)
            if (!fix_output_line(P, i, false) && force) fix_output_line(P, i, true);
    }
    // a temporary whose line starts with `-` is negated together with its uses when fewer lines start with `-` (:926-966)
    void negate_temporaries(Prog &P) {
        for (auto &L : P) lead_with_plus_group(L);
        size_t minus = leading_minus_lines(P);
        if (!minus) return;
        for (size_t i = 0; i < P.size(); ++i) {
            if (P[i][2] != W_MINUS || W.first(P[i][0]) == outchar) continue;
            std::map<size_t, Line> changed;
            changed[i] = negated(P[i]);
            const int var = P[i][0];
            while (++i < P.size()) {
                Line L(P[i]);
                if (negate_uses(L, var)) { lead_with_plus_group(L); changed[i].swap(L); }
                if (P[i][0] == var) break;
            }
            size_t now = minus;
            for (auto &kl : changed) {
                const bool was = P[kl.first][2] == W_MINUS, is = at(kl.second, 2) == W_MINUS;
                if (was && !is) --now;
                if (is && !was) ++now;
            }
            if (now < minus) { for (auto &kl : changed) P[kl.first].swap(kl.second); minus = now; }
        }
    }

    // ---- parentheses --------------------------------------------------------------------
    // end of the group that starts at `s` inside [s, e): behind the matching `)` up to the next sign / `)` / `;`, or for a
    // plain word up to the next sign (:969-986)
    static size_t group_end(const Line &v, size_t s, size_t e) {
        if (v[s] != W_LP) { while (s < e && !addsub(v[s])) ++s; return s; }
        long depth = 1; size_t k = s;
        while (depth > 0) {
            do ++k; while (k < e && v[k] != W_LP && v[k] != W_RP);
            if (k >= e) throw std::runtime_error("compacter: parenthesis not closed");
            depth += v[k] == W_LP ? 1 : -1;
        }
        do ++k; while (k < e && !addsub(v[k]) && v[k] != W_RP && v[k] != W_SEMI);
        return std::min(k, e);
    }
    // `-( -a + b ...)` -> `+( a - b ...)` when the group holds a minus to trade (:994-1064)
    void open_minus_groups(Line &L) {
        Line N; bool swapped = false;
        for (size_t s = 0; s != L.size(); ++s) {
            const size_t open = std::find(L.begin() + s, L.end(), (int)W_LP) - L.begin();
            if (open == L.size()) { N.insert(N.end(), L.begin() + s, L.end()); break; }
            const size_t close = group_end(L, open, L.size());
            size_t traded = 0; Line G;
            if (open == 0) throw std::runtime_error("compacter: malformed line");
            if (L[open - 1] == W_MINUS) {
                G.assign(L.begin() + open - 1, L.begin() + close);
                G[0] = W_PLUS;
                size_t unsigned_terms = 0, k = 2;
                do {
                    if (at(G, k) == W_RP) break;
                    if (G[k] == W_MINUS) { G[k++] = W_PLUS; ++traded; }
                    else if (G[k] == W_PLUS) G[k++] = W_MINUS;
                    else ++unsigned_terms;
                    if (k >= G.size()) throw std::runtime_error("compacter: malformed line");
                    k = group_end(G, k, G.size());
                } while (k != G.size());
                if (unsigned_terms) G.insert(G.begin() + 2, W_MINUS);
                if (at(G, 2) == W_PLUS) G.erase(G.begin() + 2);
                group_lead_with_plus(G, 0);
                group_lead_with_plus(G, 1);
            }
            if (traded) { N.insert(N.end(), L.begin() + s, L.begin() + open - 1); N.insert(N.end(), G.begin(), G.end()); swapped = true; }
            else N.insert(N.end(), L.begin() + s, L.begin() + close);
            s = close - 1;
        }
        drop_plus_after_assign(N);
        if (swapped) L.swap(N);
    }
    // `+(-a+b)` -> `-(a-b)`, `-(-a+b)` -> `+(a-b)`, `+(a-b)` and `-(a-b)` at the end of their term lose the parenthesis;
    // recursive on the words behind every group head (:1075-1142)
    Line close_minus_groups(const Line &v) {
        Line N;
        const size_t e = v.size();
        for (size_t s = 0; s != e;) {
            const size_t open = std::find(v.begin() + s, v.end(), (int)W_LP) - v.begin();
            N.insert(N.end(), v.begin() + s, v.begin() + open);
            if (open == e) break;
            const size_t close = group_end(v, open, e);
            size_t head = 1;
            if (open != s) { head = 2; N.pop_back(); }
            Line G(v.begin() + open - (head - 1), v.begin() + close);
            if (open != s && at(G, head) == W_MINUS) {
                if (addsub(G[0])) { G[0] = flipped(G[0]); G = negated(G); }
                else group_lead_with_plus(G, head - 1);
            } else if (G.back() == W_RP && at(G, 1) == W_LP) {
                if (G[0] == W_MINUS) {
                    G = negated(G);
                    const size_t keep = addsub(at(G, 2)) ? 0 : 1;
                    G.pop_back();
                    G.erase(G.begin() + keep, G.begin() + 2);
                }
                if (G[0] == W_PLUS || G[0] == W_ASSIGN) { G.pop_back(); G.erase(G.begin() + 1); }
            }
            if (G.size() < head) throw std::runtime_error("compacter: malformed line");
            N.insert(N.end(), G.begin(), G.begin() + head);
            const Line tail = close_minus_groups(Line(G.begin() + head, G.end()));
            N.insert(N.end(), tail.begin(), tail.end());
            s = close;
        }
        drop_plus_after_assign(N);
        return N;
    }

    // ---- constants ------------------------------------------------------------------------
    typedef unsigned __int128 u128;
    static bool number(const std::string &s, u128 &n, u128 &d) {       // "n" or "n/d"
        n = 0; d = 1; u128 *cur = &n; bool any = false, slash = false;
        for (unsigned char c : s) {
            if (c == '/') { if (slash || !any) return false; slash = true; cur = &d; d = 0; any = false; continue; }
            if (!isdigit(c)) return false;
            if (*cur > ((~(u128)0) >> 1) / 10 - 1) throw std::runtime_error("compacter: constant too large");
            *cur = *cur * 10 + (c - '0'); any = true;
        }
        return any && d != 0;
    }
    static u128 gcd(u128 a, u128 b) { while (b) { u128 t = a % b; a = b; b = t; } return a; }
    static u128 mul(u128 a, u128 b) { if (a && b > ((~(u128)0) >> 1) / a) throw std::runtime_error("compacter: constant product too large"); return a * b; }
    static std::string dec(u128 x) { if (!x) return "0"; std::string s; while (x) { s.push_back('0' + (int)(x % 10)); x /= 10; } std::reverse(s.begin(), s.end()); return s; }
    // `op x op y` with constants x, y becomes one factor (:1361-1405)
    void combine_constants(Line &L) {
        for (size_t k = 0; k + 4 < L.size(); ++k) {
            if (!muldiv(L[k]) || !muldiv(L[k + 2])) continue;
            u128 n, d, n2, d2;
            if (!number(W.text[L[k + 1]], n, d) || !number(W.text[L[k + 3]], n2, d2)) continue;      // not pinned: left alone
            if (n == 0 || n2 == 0) throw std::runtime_error("compacter: zero factor");
            if (L[k] == W_DIV) std::swap(n, d);
            if (L[k + 2] == W_DIV) std::swap(n2, d2);
            u128 g = gcd(n, d2), g2 = gcd(n2, d);
            n = mul(n / g, n2 / g2); d = mul(d / g2, d2 / g);
            g = gcd(n, d); n /= g; d /= g;
            if (d == 1 && n == 1) { L.erase(L.begin() + k, L.begin() + k + 4); continue; }
            if (d == 1) { L[k] = W_MUL; L[k + 1] = W.get(dec(n)); }
            else if (n == 1) { L[k] = W_DIV; L[k + 1] = W.get(dec(d)); }
            else { L[k] = W_MUL; L[k + 1] = W.get(dec(n) + "/" + dec(d)); }
            L.erase(L.begin() + k + 2, L.begin() + k + 4);
        }
    }

    // ---- one round (variablesTrimer) ---------------------------------------------------------
    char free_letter(const std::set<char> &used, char after) const {       // :246-259
        if (used.size() > 50) throw std::runtime_error("not enough free single char variables.");
        char t = after;
        do ++t; while (used.count(t) || t == 'c');
        if (t > 'z') for (t = 'A'; used.count(t); ++t) {}
        return t;
    }
    static void replace_word(Line &L, size_t from, int what, int by) { for (size_t k = from; k < L.size(); ++k) if (L[k] == what) L[k] = by; }

    void round(Prog &P, bool singles) {
        std::set<char> used;
        for (auto &L : P) for (int w : L) used.insert(W.first(w));
        const char zchar = free_letter(used, 'a' - 1), tchar = free_letter(used, zchar);

        // [1]
        {
            Prog tail;
            for (size_t i = 0; i < P.size(); ++i) {
                const int out = at(P[i], 0);
                if (W.first(out) != outchar) continue;
                std::string z = W.text[out]; z[0] = zchar;
                const int rep = W.get(z);
                for (size_t k = i; k < P.size(); ++k) replace_word(P[k], 0, out, rep);
                tail.push_back(Line{out, W_ASSIGN, rep, W_SEMI});
            }
            P.insert(P.end(), tail.begin(), tail.end());
        }
        // [2]
        for (size_t i = 0; i < P.size(); ++i) {
            if (P[i].size() != 4 || W.first(P[i][0]) == outchar) continue;
            const int x = P[i][0], a = P[i][2];
            for (size_t k = i + 1; k < P.size(); ++k) { replace_word(P[k], 2, x, a); if (P[k][0] == x) break; }
            P[i][0] = a;
        }
        P.erase(std::remove_if(P.begin(), P.end(), copy_line), P.end());
        // [3]
        for (size_t i = P.size(); i-- > 0;) {
            Line &L = P[i];
            if (L.size() != 4 || W.text[L[2]] == "0" || W.first(L[2]) == inchar || W.first(L[2]) == outchar) continue;
            for (size_t k = i; k-- > 0;) {
                if (P[k][0] == L[2]) { P[k][0] = L[0]; L[2] = L[0]; break; }
                replace_word(P[k], 2, L[2], L[0]);
            }
        }
        P.erase(std::remove_if(P.begin(), P.end(), copy_line), P.end());
        for (auto &L : P) lead_with_plus_group(L);
        fix_output_lines(P, false);
        if (!singles) return;

        // [4] every temporary gets ONE assignment (later ones are renamed), then those read once are inlined in the
        // order of their assignments, on the line numbers taken BEFORE any of it (a line emptied by an earlier
        // inlining is found empty by a later one: that one waits for the next round, as in the reference)
        size_t renamed = 0;
        std::unordered_map<int, std::vector<size_t>> seen;            // word -> lines where it stands, from its first assignment on
        std::vector<int> order;                                       // temporaries by first assignment
        for (size_t i = 0; i < P.size(); ++i) {
            Line &L = P[i];
            const int old = at(L, 0);
            if (W.first(old) != outchar && seen.count(old)) {
                L[0] = W.get(std::string(1, tchar) + std::to_string(++renamed));
                for (size_t k = i + 1; k < P.size(); ++k) replace_word(P[k], 0, old, L[0]);
            }
            if (!seen.count(L[0])) { seen[L[0]]; order.push_back(L[0]); }
            for (int w : L) { auto it = seen.find(w); if (it != seen.end()) it->second.push_back(i); }
        }
        for (int var : order) {
            const std::vector<size_t> &occ = seen[var];
            if (W.first(var) == outchar || occ.size() != 2) continue;
            if (occ[0] == occ[1]) throw std::runtime_error("compacter: variable read in its own first assignment");
            Line &def = P[occ[0]], &use = P[occ[1]];
            lead_with_plus_group(def);
            size_t at_ = 2; while (at_ < use.size() && use[at_] != var) ++at_;
            if (at_ >= use.size()) continue;
            if (at(def, 2) == W_MINUS && addsub(use[at_ - 1])) {                  // the definition's minus goes into the use's sign
                use[at_ - 1] = flipped(use[at_ - 1]);
                if (drop_plus_after_assign(use)) --at_;
                def = negated(def);
            }
            auto is_sum = [](const Line &d) { long depth = 0; for (size_t k = 3; k + 1 < d.size(); ++k) { if (d[k] == W_LP) ++depth; else if (d[k] == W_RP) --depth; else if (addsub(d[k]) && depth == 0) return true; } return false; };
            const bool sum = is_sum(def);
            bool minus = use[at_ - 1] == W_MINUS;
            const bool factor = muldiv(at(use, at_ + 1));
            if (!factor && minus && sum) {                                        // `- (a - b)` is written `+ b - a`: no parenthesis
                def = negated(def);
                lead_with_plus_group(def);
                if (at_ == 3 || def[2] == W_MINUS) use.erase(use.begin() + --at_);
                else { use[at_ - 1] = W_PLUS; if (drop_plus_after_assign(use)) --at_; }
                minus = false;
            }
            const size_t len = def.size() - 3;                                    // words of the definition's right-hand side
            use[at_] = def[2];
            use.insert(use.begin() + at_ + 1, def.begin() + 3, def.end() - 1);
            if (sum && (factor || minus)) { use.insert(use.begin() + at_, W_LP); use.insert(use.begin() + at_ + len + 1, W_RP); }
            combine_constants(use);
            def.clear();
        }
        P.erase(std::remove_if(P.begin(), P.end(), [](const Line &l) { return l.empty(); }), P.end());

        // [5]
        for (auto &L : P) {
            L = close_minus_groups(L);
            if (at(L, 1) == W_ASSIGN && at(L, 2) == W_LP && at(L, 3) == W_MINUS) { L.insert(L.begin() + 2, W_PLUS); L = close_minus_groups(L); }
            lead_with_plus_group(L);
        }
        if (leading_minus_lines(P)) { for (auto &L : P) open_minus_groups(L); negate_temporaries(P); }
        fix_output_lines(P, true);
        for (auto &L : P) lead_with_plus_group(L);
    }

    // all rounds (Compacter, src/compacter.cpp:44-58): at least two, then while the program shrinks (or `loops` of them)
    void run(Prog &P, bool singles, size_t loops, std::ostream *log) {
        round(P, singles);
        size_t now = elements(P), before;
        long left = (long)loops;
        do {
            before = now;
            round(P, singles);
            now = elements(P);
            if (log) *log << "# " << now << "\telements\tinstead of " << before << std::endl;
        } while (now < before && --left != 0);
    }
};
} } // namespace plo::trim
#endif
