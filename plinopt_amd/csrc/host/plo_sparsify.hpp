// ===========================================================================
// plo_sparsify.hpp -- host side of the change-of-basis (CoB) search of
// bin/sparsifier (reference include/plinopt_sparsify.inl, src/sparsifier.cpp),
// dependency-free: what LinBox provides there (rank, nullspace, QLUP, inverse,
// dense products) is a few dozen lines of exact Gaussian elimination here,
// because every matrix on this path is tiny (column blocks of width <= 4).
//
// The hot loop -- `localSparsifier`'s exhaustive enumeration of |Coeffs|^4
// candidate rows through `testLinComb` (:167-197, :299-314) -- is behind a
// backend interface: the host loop below, or the GPU (plo_cob_search of
// include/plinopt_hip.h) when the field is Z_p.
//
// Not specified by the reference tree (LinBox internals), hence chosen here and
// documented in DESIGN.md: the pivot rule of the LU steps (plo::sparse_lu), the
// scaling of the nullspace seed vector (free variable = 1), the order of rows of
// equal density in the seed computation (stable).  Every result is certified the
// way the reference certifies it: M == Res . CoB (`consistency`, :872-907).
// ===========================================================================
#ifndef PLO_SPARSIFY_HPP
#define PLO_SPARSIFY_HPP

#include "plo_host.hpp"

namespace plo {

template <class E> using DMat = std::vector<std::vector<E>>;     // dense, row major

template <class F> DMat<typename F::Elt> dzeros(const F &f, size_t r, size_t c) { return DMat<typename F::Elt>(r, std::vector<typename F::Elt>(c, f.zero())); }
template <class F> DMat<typename F::Elt> didentity(const F &f, size_t n) { auto I = dzeros(f, n, n); for (size_t i = 0; i < n; ++i) I[i][i] = f.one(); return I; }
template <class F> DMat<typename F::Elt> dtranspose(const F &f, const DMat<typename F::Elt> &A) {
    const size_t r = A.size(), c = r ? A[0].size() : 0; auto T = dzeros(f, c, r);
    for (size_t i = 0; i < r; ++i) for (size_t j = 0; j < c; ++j) T[j][i] = A[i][j];
    return T;
}
template <class F> DMat<typename F::Elt> dmul(const F &f, const DMat<typename F::Elt> &A, const DMat<typename F::Elt> &B) {
    const size_t r = A.size(), k = B.size(), c = k ? B[0].size() : 0; auto C = dzeros(f, r, c);
    for (size_t i = 0; i < r; ++i) for (size_t t = 0; t < k; ++t) { if (f.isZero(A[i][t])) continue; for (size_t j = 0; j < c; ++j) if (!f.isZero(B[t][j])) C[i][j] = f.add(C[i][j], f.mul(A[i][t], B[t][j])); }
    return C;
}
template <class F> size_t ddensity(const F &f, const DMat<typename F::Elt> &A) { size_t s = 0; for (auto &r : A) for (auto &e : r) if (!f.isZero(e)) ++s; return s; }
template <class F> DMat<typename F::Elt> to_dense(const F &f, const SparseMat<typename F::Elt> &M) {
    auto D = dzeros(f, M.rowdim(), M.coldim());
    for (size_t i = 0; i < M.rowdim(); ++i) for (auto &e : M.rows[i]) D[i][e.first] = e.second;
    return D;
}
template <class F> SparseMat<typename F::Elt> to_sparse(const F &f, const DMat<typename F::Elt> &D) {
    SparseMat<typename F::Elt> M(D.size(), D.empty() ? 0 : D[0].size());
    for (size_t i = 0; i < D.size(); ++i) for (size_t j = 0; j < D[i].size(); ++j) if (!f.isZero(D[i][j])) M.rows[i].emplace_back(j, D[i][j]);
    return M;
}

// reduced row echelon form in place; returns the pivot columns
template <class F> std::vector<size_t> rref(const F &f, DMat<typename F::Elt> &A) {
    std::vector<size_t> piv; const size_t r = A.size(), c = r ? A[0].size() : 0; size_t row = 0;
    for (size_t col = 0; col < c && row < r; ++col) {
        size_t p = row; while (p < r && f.isZero(A[p][col])) ++p;
        if (p == r) continue;
        std::swap(A[p], A[row]);
        const auto iv = f.inv(A[row][col]);
        for (size_t j = col; j < c; ++j) A[row][j] = f.mul(A[row][j], iv);
        for (size_t i = 0; i < r; ++i) if (i != row && !f.isZero(A[i][col])) {
            const auto l = A[i][col];
            for (size_t j = col; j < c; ++j) A[i][j] = f.add(A[i][j], f.neg(f.mul(l, A[row][j])));
        }
        piv.push_back(col); ++row;
    }
    return piv;
}
template <class F> size_t drank(const F &f, DMat<typename F::Elt> A) { return rref(f, A).size(); }
// basis of the right nullspace {x : A x = 0}, one vector per free column (free variable = 1)
template <class F> DMat<typename F::Elt> nullspace(const F &f, DMat<typename F::Elt> A, size_t ncols) {
    const auto piv = rref(f, A);
    std::vector<char> isp(ncols, 0); for (size_t c : piv) isp[c] = 1;
    DMat<typename F::Elt> N;
    for (size_t fc = 0; fc < ncols; ++fc) {
        if (isp[fc]) continue;
        std::vector<typename F::Elt> x(ncols, f.zero()); x[fc] = f.one();
        for (size_t k = 0; k < piv.size(); ++k) x[piv[k]] = f.neg(A[k][fc]);
        N.push_back(x);
    }
    return N;
}
template <class F> DMat<typename F::Elt> dinverse(const F &f, const DMat<typename F::Elt> &A) {
    const size_t n = A.size(); auto W = dzeros(f, n, 2 * n);
    for (size_t i = 0; i < n; ++i) { for (size_t j = 0; j < n; ++j) W[i][j] = A[i][j]; W[i][n + i] = f.one(); }
    const auto piv = rref(f, W);
    if (piv.size() != n || piv.back() != n - 1) throw std::domain_error("sparsifier: singular change of basis");
    auto I = dzeros(f, n, n);
    for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < n; ++j) I[i][j] = W[i][n + j];
    return I;
}

// ---------------------------------------------------------------------------
// One (block, row) enumeration of localSparsifier (:282-314): among the rows w whose only non-zero
// positions are the <= 4 columns of the block, with coefficients (Coeffs[i],Coeffs[j],Coeffs[k],Coeffs[l]) in
// lexicographic (i,j,k,l) order, the first one that is independent of the rows already chosen (Cand) and
// maximises (zeros(TM^T w), zeros(w)), provided it beats (w0,w1) strictly (testLinComb :167-197).
struct CobBest { int zv = -1, zw = -1; uint64_t index = 0; bool found = false; };
template <class F> struct CobBackend {
    virtual ~CobBackend() {}
    virtual CobBest best(const F &f, const DMat<typename F::Elt> &TM, const DMat<typename F::Elt> &Cand, size_t row, size_t offsetblock,
                         const std::vector<typename F::Elt> &coeffs, int w0, int w1) = 0;
    uint64_t candidates = 0; double seconds = 0;
};
template <class F> struct CobHostBackend : CobBackend<F> {
    using E = typename F::Elt;
    CobBest best(const F &f, const DMat<E> &TM, const DMat<E> &Cand, size_t row, size_t offsetblock, const std::vector<E> &coeffs, int w0, int w1) override {
        const size_t n = TM.size(), m = n ? TM[0].size() : 0, C = coeffs.size();
        // independence of w from the chosen rows <=> w . N != 0 for a basis N of their right nullspace (rank is
        // implementation independent; the reference copies Cand and eliminates it once per candidate, :172-175)
        DMat<E> prev(Cand.begin(), Cand.begin() + row);
        CobBest b; b.zv = w0; b.zw = w1;
        if (drank(f, prev) != row) { this->candidates += (uint64_t)C * C * C * C; return b; }   // dependent chosen rows: rank can never exceed `row` (:174)
        const DMat<E> N = nullspace(f, prev, n);
        std::vector<E> w(n, f.zero());
        for (size_t i = 0; i < C; ++i) for (size_t j = 0; j < C; ++j) for (size_t k = 0; k < C; ++k) for (size_t l = 0; l < C; ++l) {
            const E cf[4] = {coeffs[i], coeffs[j], coeffs[k], coeffs[l]};
            for (size_t t = 0; t < 4; ++t) if (offsetblock + t < n) w[offsetblock + t] = cf[t];
            bool indep = false;
            for (auto &x : N) { E s = f.zero(); for (size_t t = 0; t < 4 && offsetblock + t < n; ++t) s = f.add(s, f.mul(w[offsetblock + t], x[offsetblock + t])); if (!f.isZero(s)) { indep = true; break; } }
            ++this->candidates;
            if (!indep) continue;
            int zv = 0, zw = (int)n;
            for (size_t t = 0; t < 4 && offsetblock + t < n; ++t) if (!f.isZero(w[offsetblock + t])) --zw;
            for (size_t c = 0; c < m; ++c) { E s = f.zero(); for (size_t t = 0; t < 4 && offsetblock + t < n; ++t) if (!f.isZero(w[offsetblock + t])) s = f.add(s, f.mul(w[offsetblock + t], TM[offsetblock + t][c])); if (f.isZero(s)) ++zv; }
            if (zv > b.zv || (zv == b.zv && zw > b.zw)) { b.zv = zv; b.zw = zw; b.index = ((i * C + j) * C + k) * C + l; b.found = true; }
        }
        return b;
    }
};

template <class F> class Sparsifier {
    using E = typename F::Elt; using M_ = DMat<E>;
    const F &f; CobBackend<F> &backend; std::ostream &log;

    void profile(const char *tag, const M_ &A) const {
        size_t s = 0; log << "# " << tag; for (auto &r : A) { size_t k = 0; for (auto &e : r) if (!f.isZero(e)) ++k; s += k; log << k << ' '; } log << '=' << s << std::endl;
    }
    // Coefficient set of localSparsifier (:256-268): {0,1,-1}, then r,-r,1/r,-1/r for every entry r of TM not yet
    // listed (`augment` :21-35), then for i = 2,3,..., truncated to maxnumcoeff.  Over Modular<Integer> the reference
    // compares and negates the elements as plain integers (`-r`, `Element(i)` are not reduced; only 1/r and its
    // opposite are), so the list may hold the same residue several times and the loop over i always makes progress:
    // the raw integers are tracked here to reproduce exactly that list.
    std::vector<E> build_coeffs(const M_ &TM, size_t maxnumcoeff) const {
        std::vector<E> out;
        if constexpr (std::is_same<F, ZpField>::value) {
            const int64_t p = f.p;
            std::vector<int64_t> raw{0, 1, -1};
            auto aug = [&](int64_t r) {
                if (std::find(raw.begin(), raw.end(), r) != raw.end()) return;
                const int64_t red = ((r % p) + p) % p;
                if (red == 0) { raw.push_back(r); raw.push_back(-r); raw.push_back(0); raw.push_back(0); return; }   // not invertible: keeps the list growing
                const int64_t t = f.inv((uint32_t)red);
                raw.push_back(r); raw.push_back(-r); raw.push_back(t); raw.push_back(t ? p - t : 0);
            };
            for (auto &row : TM) for (auto &e : row) if (!f.isZero(e)) aug((int64_t)e);
            for (int64_t i = 2; raw.size() < maxnumcoeff; ++i) aug(i);
            if (raw.size() > maxnumcoeff) raw.resize(maxnumcoeff);
            for (int64_t r : raw) out.push_back((uint32_t)(((r % p) + p) % p));
        } else {
            out = {f.zero(), f.one(), f.mone()};
            auto aug = [&](const E &r) {
                if (std::find(out.begin(), out.end(), r) != out.end()) return;
                out.push_back(r); out.push_back(f.neg(r)); const E t = f.inv(r); out.push_back(t); out.push_back(f.neg(t));
            };
            for (auto &row : TM) for (auto &e : row) if (!f.isZero(e)) aug(e);
            for (long i = 2; out.size() < maxnumcoeff; ++i) aug(f.fromInt(i));
            if (out.size() > maxnumcoeff) out.resize(maxnumcoeff);
        }
        return out;
    }
public:
    Sparsifier(const F &ff, CobBackend<F> &b, std::ostream &lg) : f(ff), backend(b), log(lg) {}

    // FactorDiagonals :354-375: divide row i of TM and of TCoB by the most frequent value of TM's row
    void factor_diagonals(M_ &TCoB, M_ &TM) const {
        for (size_t i = 0; i < TM.size(); ++i) {
            std::map<E, int, std::function<bool(const E &, const E &)>> count([this](const E &a, const E &b) { return f.less(a, b); });
            for (auto &e : TM[i]) if (!f.isZero(e)) ++count[e];
            if (count.empty()) continue;
            auto best = count.begin(); for (auto it = count.begin(); it != count.end(); ++it) if (it->second > best->second) best = it;
            const E r = best->first;
            if (f.isOne(r)) continue;
            const E ir = f.inv(r);
            for (auto &e : TM[i]) e = f.mul(e, ir);
            for (auto &e : TCoB[i]) e = f.mul(e, ir);
        }
    }

    // localSparsifier :206-347
    void local_sparsifier(M_ &TCoB, M_ &TM, size_t maxnumcoeff) {
        const size_t n = TM.size(), m = n ? TM[0].size() : 0;
        M_ LCoB = dzeros(f, n, n);
        int cnHw = -1, rnHw = -1;
        if (n > 1) {                                                                              // :227-252 nullspace seed
            M_ N = dtranspose(f, TM);                                                              // m x n
            std::stable_sort(N.begin(), N.end(), [this](const std::vector<E> &a, const std::vector<E> &b) {
                size_t ka = 0, kb = 0; for (auto &e : a) if (!f.isZero(e)) ++ka; for (auto &e : b) if (!f.isZero(e)) ++kb; return ka > kb; });
            while (!N.empty() && drank(f, N) == n) N.pop_back();
            if (!N.empty()) {
                const M_ ns = nullspace(f, N, n);
                if (!ns.empty()) {
                    LCoB[0] = ns[0];
                    cnHw = 0; for (auto &e : LCoB[0]) if (!f.isZero(e)) ++cnHw;                      // LCoB[0].size() of the sparse row
                    rnHw = 0;
                    for (size_t c = 0; c < m; ++c) { E s = f.zero(); for (size_t i = 0; i < n; ++i) s = f.add(s, f.mul(LCoB[0][i], TM[i][c])); if (f.isZero(s)) ++rnHw; }
                }
            }
        }
        const std::vector<E> Coeffs = build_coeffs(TM, maxnumcoeff);                               // :256-268
        log << "# [SPRF] linear combination coefficients: [";
        for (size_t k = 0; k < Coeffs.size(); ++k) { if (k) log << ' '; f.write(log, Coeffs[k]); }
        log << ']' << std::endl;
        const size_t numlargeblocks = n >> 2, lastblock = n - (numlargeblocks << 2), numblocks = lastblock ? numlargeblocks + 1 : numlargeblocks;
        for (size_t block = 0; block < numblocks; ++block) {                                       // :282-328
            const size_t off = block << 2, first = std::min<size_t>(4, n - off);
            for (size_t num = 0; num < first; ++num) {
                const size_t row = num + off;
                int w0 = -1, w1 = -1; bool found = (block == 0 && num == 0);
                if (found) { w0 = rnHw; w1 = cnHw; }
                const CobBest b = backend.best(f, TM, LCoB, row, off, Coeffs, w0, w1);
                if (b.found) {
                    found = true;
                    const size_t C = Coeffs.size(); uint64_t x = b.index; size_t id[4];
                    for (int t = 3; t >= 0; --t) { id[t] = (size_t)(x % C); x /= C; }
                    std::fill(LCoB[row].begin(), LCoB[row].end(), f.zero());
                    for (size_t t = 0; t < 4 && off + t < n; ++t) LCoB[row][off + t] = Coeffs[id[t]];
                }
                for (size_t pp = 0; !found; ++pp) {                                                // :317-326 canonical fallback
                    if (pp >= n) throw std::logic_error("sparsifier: no independent canonical vector");
                    M_ A = LCoB; std::fill(A[row].begin(), A[row].end(), f.zero()); A[row][pp] = f.one();
                    if (drank(f, A) > row) { LCoB[row] = A[row]; found = true; }
                }
            }
        }
        TM = dmul(f, LCoB, TM);                                                                    // :336-344
        TCoB = dmul(f, LCoB, TCoB);
    }

    // SparseFactor :474-513
    size_t sparse_factor(M_ &TICoB, M_ &TM, size_t start, size_t increment, size_t threshold) {
        size_t s2 = ddensity(f, TM), ss, numcoeffs = start;
        profile("[SpFc] Columns profile: ", TM);
        do {
            ss = s2;
            local_sparsifier(TICoB, TM, numcoeffs);
            factor_diagonals(TICoB, TM);
            s2 = ddensity(f, TM);
            profile("[SpFc] Density profile: ", TM);
            if (numcoeffs < threshold) numcoeffs += increment;
        } while (s2 < ss);
        return s2;
    }

    // sparseLU :524-566: A <- (QL)^{-1} A, QL <- Q.L, only if U.P is sparser
    bool sparse_lu_step(M_ &QL, M_ &A, size_t sparsity) const {
        const size_t m = A.size(), n = m ? A[0].size() : 0;
        const LUFactors<F> lu = sparse_lu(f, to_sparse(f, A));
        if (lu.U.nnz() >= sparsity) return false;
        M_ nA = dzeros(f, m, n), nQL = dzeros(f, m, m);
        for (size_t k = 0; k < m; ++k) for (auto &e : lu.U.rows[k]) nA[k][lu.P[e.first]] = e.second;        // U . Pm
        for (size_t i = 0; i < m; ++i) for (auto &e : lu.L.rows[lu.Q[i]]) nQL[i][e.first] = e.second;       // Qm . L
        // rows of L beyond the rank carry no unit: complete Qm.L to an invertible matrix (those rows of U are zero)
        for (size_t i = 0; i < m; ++i) if (lu.Q[i] >= lu.rank) nQL[i][lu.Q[i]] = f.one();
        A = nA; QL = nQL;
        return true;
    }
    // sparseILU :574-600
    bool sparse_ilu(M_ &TC, M_ &A, size_t sparsity) const {
        const size_t m = A.size();
        M_ QL = didentity(f, m);
        if (!sparse_lu_step(QL, A, sparsity)) return false;
        TC = dmul(f, dinverse(f, QL), TC);                                                         // TC == QL . K
        return true;
    }
    // sparseAlternate :609-661:  M (m x n)  ->  CoB (n x n), Res (m x n) with M == Res . CoB
    void sparse_alternate(M_ &CoB, M_ &Res, const M_ &M, size_t maxnumcoeff) {
        const size_t n = M.empty() ? 0 : M[0].size();
        M_ TM = dtranspose(f, M), TICoB = didentity(f, n);
        factor_diagonals(TICoB, TM);
        if (sparse_ilu(TICoB, TM, ddensity(f, TM))) { profile("[sALT] GaussLo profile: ", TICoB); profile("[sALT] GaussUp profile: ", TM); }
        sparse_factor(TICoB, TM, 3, 4, 11);                                                        // defaults, plinopt_sparsify.h:78-80
        sparse_factor(TICoB, TM, maxnumcoeff, 1, maxnumcoeff);
        CoB = dtranspose(f, dinverse(f, TICoB));
        profile("[sALT] CoBasis profile: ", CoB);
        Res = dtranspose(f, TM);
    }
    // blockSparsifier :667-748
    void block_sparsifier(M_ &CoB, M_ &Res, const M_ &M, size_t blocksize, size_t maxnumcoeff, bool initialElimination) {
        const size_t m = M.size(), n = m ? M[0].size() : 0;
        if (blocksize <= 1) { sparse_alternate(CoB, Res, M, maxnumcoeff); return; }
        M_ U, L; bool reduced = initialElimination;
        if (initialElimination) {
            U = dtranspose(f, M); L = didentity(f, n);
            reduced = sparse_lu_step(L, U, ddensity(f, U));
            profile("[bSpr] IGaussL profile: ", L); profile("[bSpr] IGaussU profile: ", U);
        }
        const M_ A = reduced ? dtranspose(f, U) : M;
        std::vector<M_> vA, vC, vR;
        for (size_t c0 = 0; c0 < n; c0 += blocksize) {                                             // separateColumnBlocks :89-117
            const size_t bw = std::min(blocksize, n - c0); M_ blk = dzeros(f, m, bw);
            for (size_t i = 0; i < m; ++i) for (size_t j = 0; j < bw; ++j) blk[i][j] = A[i][c0 + j];
            vA.push_back(blk);
        }
        for (auto &blk : vA) { M_ C, R; sparse_alternate(C, R, blk, maxnumcoeff); vC.push_back(C); vR.push_back(R); }
        Res = dzeros(f, m, n);
        { size_t c0 = 0; for (auto &R : vR) { for (size_t i = 0; i < m; ++i) for (size_t j = 0; j < R[i].size(); ++j) Res[i][c0 + j] = R[i][j]; c0 += R.empty() ? 0 : R[0].size(); } }
        if (reduced) {                                                                             // CoB^T = [ L_blk . C_blk^T ... ]
            M_ TCoB = dzeros(f, n, n); size_t c0 = 0;
            for (size_t b = 0; b < vC.size(); ++b) {
                const size_t bw = vC[b].size(); M_ Lb = dzeros(f, n, bw);
                for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < bw; ++j) Lb[i][j] = L[i][c0 + j];
                const M_ B = dmul(f, Lb, dtranspose(f, vC[b]));
                for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < bw; ++j) TCoB[i][c0 + j] = B[i][j];
                c0 += bw;
            }
            CoB = dtranspose(f, TCoB);
        } else {                                                                                   // diagonalMatrix :49-63
            CoB = dzeros(f, n, n); size_t c0 = 0;
            for (auto &C : vC) { for (size_t i = 0; i < C.size(); ++i) for (size_t j = 0; j < C.size(); ++j) CoB[c0 + i][c0 + j] = C[i][j]; c0 += C.size(); }
        }
    }
    // consistency :872-907
    bool consistent(const M_ &M, const M_ &R, const M_ &C) const {
        const M_ A = dmul(f, R, C);
        for (size_t i = 0; i < M.size(); ++i) for (size_t j = 0; j < M[i].size(); ++j) if (!(A[i][j] == M[i][j])) return false;
        return true;
    }
};

} // namespace plo
#endif
