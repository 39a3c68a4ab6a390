// ==========================================================================
// plo_inplace.hpp -- host side of the in-place trilinear search (bin/trilplacer):
// builds, simplifies and prints the program of one candidate over the rationals.
// The GPU (plo_tril_search) only counts; the winning (seed, variant) is replayed
// here to obtain the text.  Inputs the device path refuses (entries other than
// +-1, empty rows) are searched here with OpenMP.
//
// Follows reference include/plinopt_inplace.inl: Atom/cumulate :15-124, complexity
// :133-144, orientindex/nextindex :179-236, simplify :243-311, pushvariables
// :322-393, LinearAlgorithm :400-502, TransposedDoubleAlgorithm :507-598 (-e),
// DoubleExpand :676-716, TriLinearProgram :732-806,
// SearchTriLinearAlgorithm :812-929; output syntax plinopt_inplace.h:101-112.
// Random choices: the per-candidate stream of include/plinopt_hip.h.
// ==========================================================================
#pragma once
#include "plo_host.hpp"
#include <algorithm>
#include <array>
#include <sstream>

namespace plo {

using Tricount = std::array<size_t, 3>;        // ADD, SCA, MUL

struct InAtom {
    char var; size_t src; char ope; Rat val; long des;
    bool sameops(const InAtom &p) const { return var == p.var && src == p.src && des == p.des; }
};
inline bool in_addsub(char c) { return c == '+' || c == '-'; }
inline bool in_muldiv(char c) { return c == '*' || c == '/'; }
inline char in_swap(char c) { return c == '+' ? '-' : '+'; }
inline char in_inv(char c) { return c == '*' ? '/' : '*'; }

class InplaceProgram {
    QField Q;
public:
    std::vector<InAtom> at;

    bool noop(const InAtom &a) const { return (in_addsub(a.ope) && Q.isZero(a.val)) || (in_muldiv(a.ope) && Q.isOne(a.val)); }
    bool cumulate(InAtom &t, const InAtom &p) const {
        if (!t.sameops(p)) return false;
        if (in_addsub(t.ope) && in_addsub(p.ope)) {
            t.val = (t.ope == p.ope) ? Q.add(t.val, p.val) : Q.add(t.val, Q.neg(p.val));
            if (Q.sign(t.val) < 0) { t.ope = in_swap(t.ope); t.val = Q.neg(t.val); }
            return true;
        }
        if (in_muldiv(t.ope) && in_muldiv(p.ope)) {
            t.val = (t.ope == p.ope) ? Q.mul(t.val, p.val) : Q.div(t.val, p.val);
            if (Q.abs(t.val) < Q.one()) { t.ope = in_inv(t.ope); t.val = Q.inv(t.val); }
            return true;
        }
        return false;
    }
    Tricount complexity() const {
        Tricount n{0, 0, 0};
        for (const auto &a : at) {
            if (in_addsub(a.ope)) { ++n[0]; if (!absOne(Q, a.val)) ++n[1]; }
            if (in_muldiv(a.ope)) ++n[1];
            if (a.ope == ' ') ++n[2];
        }
        return n;
    }
    bool simplify(bool transposed) {
        for (size_t i = 0; i < at.size(); ++i) {
            if (at[i].ope == ' ') continue;
            for (size_t k = i + 1; k < at.size(); ++k) {
                const InAtom &it = at[i], &nx = at[k];
                if (nx.sameops(it)) {
                    InAtom c = it;
                    if (cumulate(c, nx)) {
                        at.erase(at.begin() + (long)k);
                        if (noop(c)) at.erase(at.begin() + (long)i); else at[i] = c;
                        return true;
                    }
                }
                bool stop = (it.src == nx.src) && (nx.ope == ' ' || (in_addsub(it.ope) && in_muldiv(nx.ope)) || (in_muldiv(it.ope) && in_addsub(nx.ope)));
                stop |= transposed ? (it.des == (long)nx.src) : (it.des == (long)nx.src && nx.ope != ' ');
                stop |= ((long)it.src == nx.des);
                if (stop) break;
            }
        }
        return false;
    }
    void pushvariables(size_t numout) {
        for (size_t i = 0; i < numout; ++i) {
            bool found = false; size_t f = 0;
            for (size_t k = 0; k < at.size(); ++k) {
                if (!found) { if (at[k].ope != ' ' && at[k].src == i) { found = true; f = k; } continue; }
                const InAtom &fx = at[f], &it = at[k];
                bool rot = false;
                if (in_addsub(fx.ope)) {
                    if (fx.des == (long)it.src) { found = false; continue; }
                    if (it.src == i) { if (fx.des == it.des) rot = true; else if (in_muldiv(it.ope)) { found = false; continue; } }
                } else {
                    if (it.des == (long)i) { found = false; continue; }
                    if (it.src == i) { if (in_muldiv(it.ope)) rot = true; else { found = false; continue; } }
                }
                if (rot) { if (f + 1 != k) std::rotate(at.begin() + (long)f, at.begin() + (long)f + 1, at.begin() + (long)k); found = false; }
            }
            if (found && f + 1 != at.size()) std::rotate(at.begin() + (long)f, at.begin() + (long)f + 1, at.end());
        }
    }
};

// the candidate's view of a matrix: row l = sign[l] * M[perm[l]]
struct InRows { const QMat *M; const std::vector<uint32_t> *perm; const std::vector<int8_t> *sgn; };

inline size_t in_nextindex(size_t preci, const std::vector<std::pair<size_t, Rat>> &L, bool oriented, CandRng &rng) {
    QField Q;
    if (!oriented) return rng.next() % (uint32_t)L.size();
    size_t nexti = L.size();
    for (size_t k = 0; k < L.size(); ++k) if (L[k].first == preci) { nexti = k; break; }
    if (nexti == L.size() || !Q.isOne(L[nexti].second)) {
        std::vector<size_t> ones;
        for (size_t k = 0; k < L.size(); ++k) if (Q.isOne(L[k].second)) ones.push_back(k);
        if (!ones.empty()) nexti = ones[rng.next() % (uint32_t)ones.size()];
    }
    return nexti != L.size() ? nexti : 0;
}

inline Tricount in_linear(InplaceProgram &P, const InRows &R, char variable, bool transposed, bool oriented, CandRng &rng) {
    QField Q;
    const QMat &M = *R.M;
    size_t preci = M.coldim();
    auto mone = [&](char op, const Rat &v) { return Q.isMOne(v) ? in_swap(op) : op; };
    for (size_t l = 0; l < M.rowdim(); ++l) {
        std::vector<std::pair<size_t, Rat>> L = M.rows[(*R.perm)[l]];
        if ((*R.sgn)[l] < 0) for (auto &e : L) e.second = Q.neg(e.second);
        if (L.empty()) { P.at.push_back({' ', l, ' ', Rat(0), -1}); continue; }
        const size_t ai = in_nextindex(preci, L, oriented, rng);
        const size_t i = L[ai].first; const Rat av = L[ai].second;
        if (!Q.isOne(av)) { if (transposed) { if (!Q.isMOne(av)) P.at.push_back({variable, i, '/', av, -1}); } else P.at.push_back({variable, i, '*', av, -1}); }
        for (size_t k = 0; k < L.size(); ++k) if (k != ai) {
            if (transposed) P.at.push_back({variable, L[k].first, mone('-', av), L[k].second, (long)i});
            else P.at.push_back({variable, i, '+', L[k].second, (long)L[k].first});
        }
        P.at.push_back({variable, i, ' ', av, -1});
        for (size_t k = 0; k < L.size(); ++k) if (k != ai) {
            if (transposed) P.at.push_back({variable, L[k].first, mone('+', av), L[k].second, (long)i});
            else P.at.push_back({variable, i, '-', L[k].second, (long)L[k].first});
        }
        if (!Q.isOne(av)) { if (transposed) { if (!Q.isMOne(av)) P.at.push_back({variable, i, '*', av, -1}); } else P.at.push_back({variable, i, '/', av, -1}); }
        if (L.size() > 1) preci = i;
    }
    P.at.erase(std::remove_if(P.at.begin(), P.at.end(), [&](const InAtom &a) { return in_muldiv(a.ope) && Q.isOne(a.val); }), P.at.end());
    bool simp;
    do { if (transposed) P.pushvariables(M.coldim()); simp = P.simplify(transposed); } while (simp);
    return P.complexity();
}

// TransposedDoubleAlgorithm :507-598 on TT = DoubleExpand(T) (:676-716: row 2l of TT is row l of T, row 2l+1 the same entries
// one column to the right; TT has one more column), built on the fly.  One trip per pair of rows (2l, 2l+1) = one block
// <<a|c>,<0|a>> and the two barriers of one MULTD (the reference's loop header reads `++l`, which would work on every odd
// row again as an upper row and leave barriers that the synchronisation loop of TriLinearProgram cannot pass: see
// oracle/plo_tril_oracle.c).
inline Tricount in_transposed_double(InplaceProgram &P, const InRows &R, char variable) {
    QField Q;
    const QMat &M = *R.M;
    auto mone = [&](char op, const Rat &v) { return Q.isMOne(v) ? in_swap(op) : op; };
    auto notabsone = [&](const Rat &v) { return !Q.isOne(v) && !Q.isMOne(v); };
    for (size_t l = 0; l < M.rowdim(); ++l) {
        std::vector<std::pair<size_t, Rat>> L = M.rows[(*R.perm)[l]];
        if ((*R.sgn)[l] < 0) for (auto &e : L) e.second = Q.neg(e.second);
        if (L.empty()) { P.at.push_back({' ', 2 * l, ' ', Rat(0), -1}); continue; }
        const size_t i = L[0].first, cindex = i + 1;
        const Rat a = L[0].second, y = Q.inv(a);
        Rat c(0), z(0);
        if (L.size() > 1 && L[1].first == cindex) { c = L[1].second; z = Q.neg(Q.mul(Q.mul(y, c), y)); }
        if (notabsone(y)) P.at.push_back({variable, cindex, '*', y, -1});
        if (!Q.isZero(z)) P.at.push_back({variable, cindex, mone('+', y), z, (long)i});
        if (notabsone(y)) P.at.push_back({variable, i, '*', y, -1});
        for (size_t k = 1; k < L.size(); ++k) {
            if (L[k].first != cindex) P.at.push_back({variable, L[k].first, mone('-', y), L[k].second, (long)i});
            P.at.push_back({variable, L[k].first + 1, mone('-', y), L[k].second, (long)cindex});
        }
        P.at.push_back({variable, i, ' ', a, -1});
        P.at.push_back({variable, cindex, ' ', a, -1});
        for (size_t k = 1; k < L.size(); ++k) {
            if (L[k].first != cindex) P.at.push_back({variable, L[k].first, mone('+', a), L[k].second, (long)i});
            P.at.push_back({variable, L[k].first + 1, mone('+', a), L[k].second, (long)cindex});
        }
        if (notabsone(a)) P.at.push_back({variable, cindex, '*', a, -1});
        if (!Q.isZero(c)) P.at.push_back({variable, cindex, mone('+', a), c, (long)i});
        if (notabsone(a)) P.at.push_back({variable, i, '*', a, -1});
    }
    P.at.erase(std::remove_if(P.at.begin(), P.at.end(), [&](const InAtom &a) { return in_muldiv(a.ope) && Q.isOne(a.val); }), P.at.end());
    bool simp;
    do { P.pushvariables(M.coldim() + 1); simp = P.simplify(true); } while (simp);
    return P.complexity();
}

inline void in_print_atom(std::ostream &os, const InAtom &p) {
    QField Q; size_t dummy = 0;
    const bool sca = in_muldiv(p.ope);
    if (sca && Q.isOne(p.val)) return;
    if (p.ope == ' ') { if (Q.isZero(p.val)) os << "0;"; else os << p.var << p.src << ';'; os << '\n'; return; }
    os << p.var << p.src << ":=";
    const Rat uval = Q.abs(p.val);
    if (sca) { if (Q.sign(p.val) < 0) os << '-'; Q.print_mul(os, p.var, p.src, p.ope == '*' ? uval : Q.inv(uval), dummy); }
    else { os << p.var << p.src << (Q.sign(p.val) < 0 ? in_swap(p.ope) : p.ope); Q.print_mul(os, p.var, (size_t)p.des, uval, dummy); }
    os << ";\n";
}

struct TrilCandidate {
    std::vector<uint32_t> perm; std::vector<int8_t> sa, sb, st;
    Tricount ops[2]; std::string text[2];
};

// one restart (:837-924): both variants; text only for `want` (0, 1, or -1 none)
inline TrilCandidate tril_candidate(const QMat &A, const QMat &B, const QMat &T, uint64_t seed, int want, bool expanded = false) {
    const size_t m = A.rowdim();
    TrilCandidate C; C.perm.resize(m); C.sa.assign(m, 1); C.sb.assign(m, 1); C.st.assign(m, 1);
    for (size_t i = 0; i < m; ++i) C.perm[i] = (uint32_t)i;
    CandRng rng(seed);
    const bool base = seed == ~0ull;
    if (!base) {
        for (size_t i = m; i > 1; --i) std::swap(C.perm[i - 1], C.perm[rng.next() % (uint32_t)i]);
        for (size_t i = 0; i < m; ++i) { const bool na = rng.next() & 1u, nb = rng.next() & 1u; if (na) C.sa[i] = -1; if (nb) C.sb[i] = -1; if (na != nb) C.st[i] = -1; }
    }
    QField Q;
    for (int variant = 0; variant < 2; ++variant) {
        if (base && variant == 1) { C.ops[1] = C.ops[0]; C.text[1] = C.text[0]; break; }
        InplaceProgram pa, pb, pc;
        const Tricount oa = in_linear(pa, InRows{&A, &C.perm, &C.sa}, 'a', false, variant == 0, rng);
        const Tricount ob = in_linear(pb, InRows{&B, &C.perm, &C.sb}, 'b', false, variant == 0, rng);
        Tricount oc = expanded ? in_transposed_double(pc, InRows{&T, &C.perm, &C.st}, 'c') : in_linear(pc, InRows{&T, &C.perm, &C.st}, 'c', true, variant == 0, rng);
        const Tricount oc_print = oc;
        if (expanded) oc[2] >>= 1;                                   // :799: MUL2D is counted twice
        C.ops[variant] = {oa[0] + ob[0] + oc[0], oa[1] + ob[1] + oc[1], (oa[2] + ob[2] + oc[2]) / 3};
        if (want == variant || (base && want >= 0)) {
            std::ostringstream os;
            auto tri = [&](const Tricount &t) { os << t[0] << '|' << t[1] << '|' << t[2]; };
            os << "# Found "; tri(oa); os << " for a\n# Found "; tri(ob); os << " for b\n# Found "; tri(oc_print); os << " for c\n";
            size_t ia = 0, ib = 0, ic = 0;
            while (ic < pc.at.size()) {
                for (; ia < pa.at.size() && pa.at[ia].ope != ' '; ++ia) in_print_atom(os, pa.at[ia]);
                for (; ib < pb.at.size() && pb.at[ib].ope != ' '; ++ib) in_print_atom(os, pb.at[ib]);
                for (; ic < pc.at.size() && pc.at[ic].ope != ' '; ++ic) in_print_atom(os, pc.at[ic]);
                if (ia < pa.at.size() && ib < pb.at.size() && ic < pc.at.size()) {
                    const InAtom &c = pc.at[ic];
                    if (expanded) {                                   // MULTD, plinopt_inplace.h:111
                        const InAtom &c2 = ic + 1 < pc.at.size() ? pc.at[ic + 1] : c;
                        os << c.var << c.src << ":=" << c.var << c.src << ' ' << (Q.isMOne(c.val) ? '-' : '+') << " (" << pa.at[ia].var << pa.at[ia].src << " * "
                           << pb.at[ib].var << pb.at[ib].src << ")*low; ### AXPY low  ###\n";
                        os << c2.var << c2.src << ":=" << c2.var << c2.src << ' ' << (Q.isMOne(c.val) ? '-' : '+') << " (" << pa.at[ia].var << pa.at[ia].src << " * "
                           << pb.at[ib].var << pb.at[ib].src << ")*hig; ### AXPY high ###\n";
                        ++ic;
                    } else
                    os << c.var << c.src << ":=" << c.var << c.src << ' ' << (Q.isMOne(c.val) ? '-' : '+') << ' ' << pa.at[ia].var << pa.at[ia].src << " * "
                       << pb.at[ib].var << pb.at[ib].src << "; ### AXPY ###\n";
                    ++ia; ++ib; ++ic;
                }
            }
            for (; ic < pc.at.size(); ++ic) in_print_atom(os, pc.at[ic]);
            for (; ib < pb.at.size(); ++ib) in_print_atom(os, pb.at[ib]);
            for (; ia < pa.at.size(); ++ia) in_print_atom(os, pa.at[ia]);
            C.text[variant] = os.str();
        }
    }
    return C;
}

} // namespace plo
