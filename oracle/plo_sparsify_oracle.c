/* ===========================================================================
 * plo_sparsify_oracle.c -- CPU restatement of the reference's change-of-basis
 * search (bin/sparsifier), whole tool, over Z_p.  TEST INFRASTRUCTURE ONLY: tests/
 * and the cpu_baseline leg of bench.py may call it, never the product.
 *
 * Restated literally from /root/reference/include/plinopt_sparsify.inl:
 *   augment                      :21-35
 *   coefficient set              :256-268  (Coeffs = {0,1,-1}, entries of TM, then 2,3,...)
 *   testLinComb                  :167-197  (rank of a COPY of the candidate matrix per candidate, :38-45)
 *   localSparsifier              :206-347  (nullspace seed :227-252, enumeration :282-314, canonical fallback :317-326,
 *                                           application of LCoB :336-344)
 *   FactorDiagonals              :355-375
 *   SparseFactor                 :474-513
 *   sparseLU / sparseILU         :524-600
 *   sparseAlternate              :610-661
 *   blockSparsifier              :667-748  (separateColumnBlocks :89-117, augmentedMatrix :67-86, diagonalMatrix :49-63)
 * Dense row-major arrays of residues, plain loops, no sharing of code with plinopt_amd/.
 *
 * NOT determined by the reference tree (LinBox internals; SURVEY.md 8c, "parity unpinned"): the build's rules, restated here
 * from their DESCRIPTION in DESIGN.md 2.5 / plo_sparsify.hpp:13-17, by a different route (dense elimination instead of maps):
 *   - QLUP (GaussDomain::QLUPin): pivot = the first row, in index order, that still has an entry and was not a pivot; in it the
 *     entry of smallest column; rows below are reduced in index order; U keeps the pivot rows in pivot order, L the multipliers;
 *   - nullspace seed (nullspacebasisin): reduced row echelon form with the first usable row as pivot of each column; ONE vector,
 *     that of the first free column, free variable = 1;
 *   - rows of equal density keep their order in the sort of :230 (stable).
 * Over Modular<Integer> the reference builds the coefficient list with unreduced integers (`-r`, `Element(i)`): the raw
 * integers are tracked, as src/sparsifier.cpp:74-83 would with Givaro::Modular<Integer>.
 * =========================================================================== */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t u32;
static u32 P_;                                                 /* the modulus of the running call (single-threaded checker) */
static u32 fadd(u32 a, u32 b) { return (u32)(((uint64_t)a + b) % P_); }
static u32 fneg(u32 a) { return a ? P_ - a : 0; }
static u32 fmul(u32 a, u32 b) { return (u32)(((uint64_t)a * b) % P_); }
static u32 finv(u32 a) { int64_t t = 0, nt = 1, r = P_, nr = a % P_; while (nr) { int64_t q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; } if (t < 0) t += P_; return (u32)t; }
static void *xmalloc(size_t n) { void *q = calloc(n ? n : 1, 1); if (!q) abort(); return q; }

/* rank of an r x c matrix by elimination of a COPY (rank :38-45) */
static u32 rank_of(const u32 *A, u32 r, u32 c)
{
    u32 *W = xmalloc(sizeof(u32) * (size_t)r * c); memcpy(W, A, sizeof(u32) * (size_t)r * c);
    u32 rk = 0;
    for (u32 col = 0; col < c && rk < r; ++col) {
        u32 p = rk; while (p < r && W[(size_t)p * c + col] == 0) ++p;
        if (p == r) continue;
        if (p != rk) for (u32 j = 0; j < c; ++j) { u32 t = W[(size_t)p * c + j]; W[(size_t)p * c + j] = W[(size_t)rk * c + j]; W[(size_t)rk * c + j] = t; }
        const u32 iv = finv(W[(size_t)rk * c + col]);
        for (u32 i = rk + 1; i < r; ++i) if (W[(size_t)i * c + col]) {
            const u32 l = fmul(W[(size_t)i * c + col], iv);
            for (u32 j = col; j < c; ++j) W[(size_t)i * c + j] = fadd(W[(size_t)i * c + j], fneg(fmul(l, W[(size_t)rk * c + j])));
        }
        ++rk;
    }
    free(W);
    return rk;
}
static void matmul(u32 *C, const u32 *A, const u32 *B, u32 r, u32 k, u32 c)     /* C (r x c) = A (r x k) . B (k x c) */
{
    for (u32 i = 0; i < r; ++i) for (u32 j = 0; j < c; ++j) { u32 s = 0; for (u32 t = 0; t < k; ++t) s = fadd(s, fmul(A[(size_t)i * k + t], B[(size_t)t * c + j])); C[(size_t)i * c + j] = s; }
}
static void transpose(u32 *T, const u32 *A, u32 r, u32 c) { for (u32 i = 0; i < r; ++i) for (u32 j = 0; j < c; ++j) T[(size_t)j * r + i] = A[(size_t)i * c + j]; }
static u32 density(const u32 *A, u32 r, u32 c) { u32 s = 0; for (size_t k = 0; k < (size_t)r * c; ++k) if (A[k]) ++s; return s; }
/* inverse of an n x n matrix (Gauss-Jordan on [A | I]); 0 when singular */
static int inverse(u32 *I, const u32 *A, u32 n)
{
    u32 *W = xmalloc(sizeof(u32) * (size_t)n * 2 * n);
    for (u32 i = 0; i < n; ++i) { for (u32 j = 0; j < n; ++j) W[(size_t)i * 2 * n + j] = A[(size_t)i * n + j]; W[(size_t)i * 2 * n + n + i] = 1 % P_; }
    for (u32 col = 0; col < n; ++col) {
        u32 p = col; while (p < n && W[(size_t)p * 2 * n + col] == 0) ++p;
        if (p == n) { free(W); return 0; }
        if (p != col) for (u32 j = 0; j < 2 * n; ++j) { u32 t = W[(size_t)p * 2 * n + j]; W[(size_t)p * 2 * n + j] = W[(size_t)col * 2 * n + j]; W[(size_t)col * 2 * n + j] = t; }
        const u32 iv = finv(W[(size_t)col * 2 * n + col]);
        for (u32 j = 0; j < 2 * n; ++j) W[(size_t)col * 2 * n + j] = fmul(W[(size_t)col * 2 * n + j], iv);
        for (u32 i = 0; i < n; ++i) if (i != col && W[(size_t)i * 2 * n + col]) {
            const u32 l = W[(size_t)i * 2 * n + col];
            for (u32 j = 0; j < 2 * n; ++j) W[(size_t)i * 2 * n + j] = fadd(W[(size_t)i * 2 * n + j], fneg(fmul(l, W[(size_t)col * 2 * n + j])));
        }
    }
    for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < n; ++j) I[(size_t)i * n + j] = W[(size_t)i * 2 * n + n + j];
    free(W);
    return 1;
}

/* ---- coefficient set :256-268 with augment :21-35 (raw integers, see the header) */
typedef struct { int64_t *raw; u32 n, cap; } coefs_t;
static int co_has(const coefs_t *c, int64_t r) { for (u32 k = 0; k < c->n; ++k) if (c->raw[k] == r) return 1; return 0; }
static void co_push(coefs_t *c, int64_t r) { if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->raw = realloc(c->raw, sizeof(int64_t) * c->cap); if (!c->raw) abort(); } c->raw[c->n++] = r; }
static void augment(coefs_t *c, int64_t r)
{
    if (co_has(c, r)) return;                                   /* std::find on the elements as stored :27 */
    int64_t red = r % (int64_t)P_; if (red < 0) red += P_;
    co_push(c, r); co_push(c, -r);                              /* v.push_back(r); v.push_back(-r) :28-29 */
    if (red == 0) { co_push(c, 0); co_push(c, 0); return; }     /* (not invertible: the list still grows by four) */
    const int64_t t = finv((u32)red);
    co_push(c, t); co_push(c, t ? (int64_t)P_ - t : 0);         /* inv, then negin :31-33 */
}
static u32 build_coeffs(u32 *out, const u32 *TM, u32 n, u32 m, u32 maxnumcoeff)
{
    coefs_t c = {0, 0, 0};
    co_push(&c, 0); co_push(&c, 1); co_push(&c, -1);            /* :256 */
    for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < m; ++j) if (TM[(size_t)i * m + j]) augment(&c, (int64_t)TM[(size_t)i * m + j]);   /* stored entries, row major :257-261 */
    for (int64_t i = 2; c.n < maxnumcoeff; ++i) augment(&c, i); /* :263-265 */
    if (c.n > maxnumcoeff) c.n = maxnumcoeff;                   /* :268 */
    for (u32 k = 0; k < c.n; ++k) { int64_t r = c.raw[k] % (int64_t)P_; if (r < 0) r += P_; out[k] = (u32)r; }
    const u32 cn = c.n; free(c.raw);
    return cn;
}

/* ---- testLinComb :167-197 (Cand: n x n, a copy whose row `num` is replaced by w) */
static int test_lin_comb(int *w0, int *w1, u32 *LCoB, u32 *Cand, u32 num, const u32 *w, const u32 *TM, u32 n, u32 m)
{
    memcpy(Cand + (size_t)num * n, w, sizeof(u32) * n);         /* setRow(Cand,num,w) :172 */
    if (rank_of(Cand, n, n) > num) {                            /* :173-175 */
        int rl = 0, cl = 0;
        for (u32 c = 0; c < m; ++c) { u32 s = 0; for (u32 i = 0; i < n; ++i) s = fadd(s, fmul(w[i], TM[(size_t)i * m + c])); if (s == 0) ++rl; }   /* applyTranspose, zeros :176-179 */
        for (u32 i = 0; i < n; ++i) if (w[i] == 0) ++cl;        /* :180 */
        if (rl > *w0 || (rl == *w0 && cl > *w1)) { *w0 = rl; *w1 = cl; memcpy(LCoB + (size_t)num * n, w, sizeof(u32) * n); return 1; }   /* :183-194 */
    }
    return 0;
}

/* ---- localSparsifier :206-347; TM: n x m, TCoB: n x n, both updated */
static uint64_t g_candidates, g_carried, g_fallbacks;      /* g_carried: candidates evaluated with a coordinate outside their block (the fallback's w[p] = 1 kept by :305) */
static void local_sparsifier(u32 *TCoB, u32 *TM, u32 n, u32 m, u32 maxnumcoeff)
{
    u32 *LCoB = xmalloc(sizeof(u32) * (size_t)n * n);
    int cnHw = -1, rnHw = -1;
    if (n > 1) {                                                /* :227-252 */
        /* N = TM^T (m x n), rows sorted by decreasing size, stable; rows dropped from the end while the rank is n */
        u32 *N = xmalloc(sizeof(u32) * (size_t)m * n), *ord = xmalloc(sizeof(u32) * m), *sz = xmalloc(sizeof(u32) * m);
        for (u32 c = 0; c < m; ++c) { ord[c] = c; sz[c] = 0; for (u32 i = 0; i < n; ++i) if (TM[(size_t)i * m + c]) ++sz[c]; }
        for (u32 a = 1; a < m; ++a) { const u32 x = ord[a]; u32 b = a; while (b > 0 && sz[ord[b - 1]] < sz[x]) { ord[b] = ord[b - 1]; --b; } ord[b] = x; }   /* insertion sort: stable */
        for (u32 k = 0; k < m; ++k) for (u32 i = 0; i < n; ++i) N[(size_t)k * n + i] = TM[(size_t)i * m + ord[k]];
        u32 rows = m;
        while (rows > 0 && rank_of(N, rows, n) == n) --rows;    /* :231-233 */
        if (rows > 0) {
            /* reduced row echelon form of the kept rows; the vector of the FIRST free column, free variable = 1 (the build's rule) */
            u32 *W = xmalloc(sizeof(u32) * (size_t)rows * n); memcpy(W, N, sizeof(u32) * (size_t)rows * n);
            u32 *pivcol = xmalloc(sizeof(u32) * n); u32 rk = 0;
            for (u32 col = 0; col < n && rk < rows; ++col) {
                u32 p = rk; while (p < rows && W[(size_t)p * n + col] == 0) ++p;
                if (p == rows) continue;
                if (p != rk) for (u32 j = 0; j < n; ++j) { u32 t = W[(size_t)p * n + j]; W[(size_t)p * n + j] = W[(size_t)rk * n + j]; W[(size_t)rk * n + j] = t; }
                const u32 iv = finv(W[(size_t)rk * n + col]);
                for (u32 j = 0; j < n; ++j) W[(size_t)rk * n + j] = fmul(W[(size_t)rk * n + j], iv);
                for (u32 i = 0; i < rows; ++i) if (i != rk && W[(size_t)i * n + col]) { const u32 l = W[(size_t)i * n + col]; for (u32 j = 0; j < n; ++j) W[(size_t)i * n + j] = fadd(W[(size_t)i * n + j], fneg(fmul(l, W[(size_t)rk * n + j]))); }
                pivcol[rk++] = col;
            }
            u32 fc = n;
            for (u32 col = 0; col < n && fc == n; ++col) { int isp = 0; for (u32 k = 0; k < rk; ++k) if (pivcol[k] == col) isp = 1; if (!isp) fc = col; }
            if (fc < n) {
                for (u32 i = 0; i < n; ++i) LCoB[i] = 0;
                LCoB[fc] = 1 % P_;
                for (u32 k = 0; k < rk; ++k) LCoB[pivcol[k]] = fneg(W[(size_t)k * n + fc]);          /* x_pivot = - entry of the free column */
                cnHw = 0; for (u32 i = 0; i < n; ++i) if (LCoB[i]) ++cnHw;                              /* LCoB[0].size() :245 */
                rnHw = 0; for (u32 c = 0; c < m; ++c) { u32 s = 0; for (u32 i = 0; i < n; ++i) s = fadd(s, fmul(LCoB[i], TM[(size_t)i * m + c])); if (s == 0) ++rnHw; }   /* :243-246 */
            }
            free(W); free(pivcol);
        }
        free(N); free(ord); free(sz);
    }
    u32 *Coeffs = xmalloc(sizeof(u32) * (maxnumcoeff + 8));
    const u32 C = build_coeffs(Coeffs, TM, n, m, maxnumcoeff);
    const u32 numlargeblocks = n >> 2, lastblock = n - (numlargeblocks << 2), numblocks = lastblock ? numlargeblocks + 1 : numlargeblocks;   /* :274-277 */
    const u32 multiple = numblocks << 2;                        /* :277 (>= n) */
    u32 *A = xmalloc(sizeof(u32) * (size_t)n * n), *w = xmalloc(sizeof(u32) * ((size_t)multiple + 4));
    for (u32 block = 0; block < numblocks; ++block) {
        memset(w, 0, sizeof(u32) * ((size_t)multiple + 4));     /* w.resize(0); w.resize(multiple) :283 -- ONCE PER BLOCK */
        const u32 off = block << 2, first = n - off < 4 ? n - off : 4;
        for (u32 num = 0; num < first; ++num) {
            memcpy(A, LCoB, sizeof(u32) * (size_t)n * n);       /* matrixCopy(A, LCoB) :288 */
            int w0 = -1, w1 = -1; int found = (block == 0 && num == 0);
            if (found) { w0 = rnHw; w1 = cnHw; }                /* :289-294 */
            for (u32 i = 0; i < C; ++i) for (u32 j = 0; j < C; ++j) for (u32 k = 0; k < C; ++k) for (u32 l = 0; l < C; ++l) {   /* :299-314 */
                /* w.resize(multiple) :305 zero-fills the positions n.. only: what w holds below n STAYS -- after the canonical
                 * fallback of an earlier row of this block (below) that is its w[p] = 1, for p outside the block */
                for (u32 x = n; x < multiple + 4; ++x) w[x] = 0;
                w[off] = Coeffs[i]; w[off + 1] = Coeffs[j]; w[off + 2] = Coeffs[k]; w[off + 3] = Coeffs[l];      /* :306-309 */
                /* w.resize(TM.rowdim()) :311: positions beyond n are dropped (test_lin_comb reads n words) */
                ++g_candidates;
                for (u32 x = 0; x < n; ++x) if (w[x] && (x < off || x >= off + 4)) { ++g_carried; break; }
                found |= test_lin_comb(&w0, &w1, LCoB, A, num + off, w, TM, n, m);
            }
            for (u32 pp = 0; !found && pp < n; ++pp) {           /* canonical fallback :317-326 */
                w0 = -1; w1 = -1;
                memset(w, 0, sizeof(u32) * ((size_t)multiple + 4)); w[pp] = 1 % P_;      /* w.resize(0); w.resize(rowdim); w[p]=1 :320-321 */
                if (pp == 0) ++g_fallbacks;
                found |= test_lin_comb(&w0, &w1, LCoB, A, num + off, w, TM, n, m);
            }
        }
    }
    u32 *TR = xmalloc(sizeof(u32) * (size_t)n * m), *TS = xmalloc(sizeof(u32) * (size_t)n * n);
    matmul(TR, LCoB, TM, n, n, m); matmul(TS, LCoB, TCoB, n, n, n);                /* :336-344 */
    memcpy(TM, TR, sizeof(u32) * (size_t)n * m); memcpy(TCoB, TS, sizeof(u32) * (size_t)n * n);
    free(LCoB); free(Coeffs); free(A); free(w); free(TR); free(TS);
}

/* ---- FactorDiagonals :355-375 (std::map by value, max_element = first of the largest counts) */
static void factor_diagonals(u32 *TCoB, u32 *TM, u32 n, u32 m)
{
    for (u32 i = 0; i < n; ++i) {
        u32 best = 0; int bestc = 0;
        for (u32 j = 0; j < m; ++j) { const u32 v = TM[(size_t)i * m + j]; if (!v) continue;
            int c = 0; for (u32 t = 0; t < m; ++t) if (TM[(size_t)i * m + t] == v) ++c;
            if (c > bestc || (c == bestc && v < best)) { bestc = c; best = v; } }                /* ascending keys, strict > keeps the smallest */
        if (!bestc || best == 1 % P_) continue;
        const u32 ir = finv(best);
        for (u32 j = 0; j < m; ++j) TM[(size_t)i * m + j] = fmul(TM[(size_t)i * m + j], ir);
        for (u32 j = 0; j < n; ++j) TCoB[(size_t)i * n + j] = fmul(TCoB[(size_t)i * n + j], ir);
    }
}

/* ---- SparseFactor :474-513 */
static u32 sparse_factor(u32 *TICoB, u32 *TM, u32 n, u32 m, u32 start, u32 increment, u32 threshold)
{
    u32 s2 = density(TM, n, m), ss, numcoeffs = start;
    do {
        ss = s2;
        local_sparsifier(TICoB, TM, n, m, numcoeffs);
        factor_diagonals(TICoB, TM, n, m);
        s2 = density(TM, n, m);
        if (numcoeffs < threshold) numcoeffs += increment;
    } while (s2 < ss);
    return s2;
}

/* ---- sparseLU :524-566 with the build's pivot rule: A (r x c) <- U.P, QL (r x r) <- Q.L, only when sparser */
static int sparse_lu(u32 *QL, u32 *A, u32 r, u32 c, u32 sparsity)
{
    u32 *W = xmalloc(sizeof(u32) * (size_t)r * c), *Lm = xmalloc(sizeof(u32) * (size_t)r * r);   /* Lm[i][k]: multiplier of row i on pivot k */
    memcpy(W, A, sizeof(u32) * (size_t)r * c);
    int *isp = xmalloc(sizeof(int) * r); u32 *prow = xmalloc(sizeof(u32) * r); u32 rk = 0;
    for (;;) {
        u32 pr = r, pc = 0;
        for (u32 i = 0; i < r && pr == r; ++i) { if (isp[i]) continue; for (u32 j = 0; j < c; ++j) if (W[(size_t)i * c + j]) { pr = i; pc = j; break; } }
        if (pr == r) break;
        isp[pr] = 1; prow[rk] = pr;
        const u32 iv = finv(W[(size_t)pr * c + pc]);
        for (u32 i = 0; i < r; ++i) {
            if (isp[i] || W[(size_t)i * c + pc] == 0) continue;
            const u32 l = fmul(W[(size_t)i * c + pc], iv);
            for (u32 j = 0; j < c; ++j) W[(size_t)i * c + j] = fadd(W[(size_t)i * c + j], fneg(fmul(l, W[(size_t)pr * c + j])));
            Lm[(size_t)i * r + rk] = l;
        }
        ++rk;
    }
    u32 dens = 0; for (u32 k = 0; k < rk; ++k) for (u32 j = 0; j < c; ++j) if (W[(size_t)prow[k] * c + j]) ++dens;
    int sparser = dens < sparsity;                              /* density(U) < sparsity :551 */
    if (sparser) {
        /* A <- the pivot rows in pivot order (zero rows behind); QL[i][k] = 1 for pivot k's own row, the multiplier for a reduced row;
           a row that never was a pivot gets a unit in its own column behind the rank (those rows of the new A are zero) */
        u32 *nA = xmalloc(sizeof(u32) * (size_t)r * c), *nQ = xmalloc(sizeof(u32) * (size_t)r * r);
        for (u32 k = 0; k < rk; ++k) { memcpy(nA + (size_t)k * c, W + (size_t)prow[k] * c, sizeof(u32) * c); nQ[(size_t)prow[k] * r + k] = 1 % P_; }
        u32 nx = rk;
        for (u32 i = 0; i < r; ++i) {
            for (u32 k = 0; k < rk; ++k) if (Lm[(size_t)i * r + k]) nQ[(size_t)i * r + k] = Lm[(size_t)i * r + k];
            if (!isp[i]) nQ[(size_t)i * r + nx++] = 1 % P_;
        }
        memcpy(A, nA, sizeof(u32) * (size_t)r * c); memcpy(QL, nQ, sizeof(u32) * (size_t)r * r);
        free(nA); free(nQ);
    }
    free(W); free(Lm); free(isp); free(prow);
    return sparser;
}
/* ---- sparseILU :574-600 */
static int sparse_ilu(u32 *TC, u32 *A, u32 r, u32 c, u32 sparsity)
{
    u32 *QL = xmalloc(sizeof(u32) * (size_t)r * r); for (u32 i = 0; i < r; ++i) QL[(size_t)i * r + i] = 1 % P_;
    const int sparser = sparse_lu(QL, A, r, c, sparsity);
    if (sparser) {
        u32 *I = xmalloc(sizeof(u32) * (size_t)r * r), *K = xmalloc(sizeof(u32) * (size_t)r * r);
        if (!inverse(I, QL, r)) abort();
        matmul(K, I, TC, r, r, r);                              /* applyInverse: TC == QL . K */
        memcpy(TC, K, sizeof(u32) * (size_t)r * r);
        free(I); free(K);
    }
    free(QL);
    return sparser;
}

/* ---- sparseAlternate :610-661: M (m x n) -> CoB (n x n), Res (m x n) */
static int sparse_alternate(u32 *CoB, u32 *Res, const u32 *M, u32 m, u32 n, u32 maxnumcoeff)
{
    u32 *TM = xmalloc(sizeof(u32) * (size_t)n * m), *TICoB = xmalloc(sizeof(u32) * (size_t)n * n);
    transpose(TM, M, m, n);
    for (u32 i = 0; i < n; ++i) TICoB[(size_t)i * n + i] = 1 % P_;
    factor_diagonals(TICoB, TM, n, m);                          /* :627 */
    sparse_ilu(TICoB, TM, n, m, density(TM, n, m));            /* :629 */
    sparse_factor(TICoB, TM, n, m, 3, 4, 11);                   /* defaults, plinopt_sparsify.h:78-80 */
    sparse_factor(TICoB, TM, n, m, maxnumcoeff, 1, maxnumcoeff);   /* :641 */
    u32 *I = xmalloc(sizeof(u32) * (size_t)n * n);
    const int ok = inverse(I, TICoB, n);
    if (ok) { transpose(CoB, I, n, n); transpose(Res, TM, n, m); }   /* inverseTranspose :646, Transpose :651 */
    free(TM); free(TICoB); free(I);
    return ok;
}

/* the coefficient set alone (S1) */
int plo_oracle_sp_coeffs(uint32_t n, uint32_t m, const uint32_t *TM, uint32_t p, uint32_t maxnumcoeff, uint32_t *out, uint32_t *ncoeffs)
{
    P_ = p; *ncoeffs = build_coeffs(out, TM, n, m, maxnumcoeff); return 0;
}
/* one localSparsifier call (S3, with seed vector and fallback): TM n x m and TCoB n x n are updated */
int plo_oracle_sp_local(uint32_t n, uint32_t m, uint32_t *TM, uint32_t *TCoB, uint32_t p, uint32_t maxnumcoeff)
{
    P_ = p; local_sparsifier(TCoB, TM, n, m, maxnumcoeff); return 0;
}
/* blockSparsifier :667-748: M (m x n, dense, row major) -> CoB (n x n), Res (m x n) with M == Res . CoB; returns 0, 1 = a singular change of basis */
int plo_oracle_sparsify(uint32_t m, uint32_t n, const uint32_t *M, uint32_t p, uint32_t blocksize, uint32_t maxnumcoeff, int initial_elimination,
                        uint32_t *CoB, uint32_t *Res, uint64_t *candidates)
{
    P_ = p; g_candidates = 0; g_carried = 0; g_fallbacks = 0;
    int rc = 0;
    if (blocksize <= 1) { rc = sparse_alternate(CoB, Res, M, m, n, maxnumcoeff) ? 0 : 1; if (candidates) *candidates = g_candidates; return rc; }
    u32 *U = xmalloc(sizeof(u32) * (size_t)n * m), *L = xmalloc(sizeof(u32) * (size_t)n * n);
    int reduced = initial_elimination;
    if (initial_elimination) {
        transpose(U, M, m, n);
        for (u32 i = 0; i < n; ++i) L[(size_t)i * n + i] = 1 % P_;
        reduced = sparse_lu(L, U, n, m, density(U, n, m));      /* :688 */
    }
    u32 *A = xmalloc(sizeof(u32) * (size_t)m * n);
    if (reduced) transpose(A, U, n, m); else memcpy(A, M, sizeof(u32) * (size_t)m * n);        /* :699 */
    memset(Res, 0, sizeof(u32) * (size_t)m * n); memset(CoB, 0, sizeof(u32) * (size_t)n * n);
    u32 *TCoB = xmalloc(sizeof(u32) * (size_t)n * n);
    for (u32 c0 = 0; c0 < n; c0 += blocksize) {                 /* separateColumnBlocks :89-117 */
        const u32 bw = n - c0 < blocksize ? n - c0 : blocksize;
        u32 *blk = xmalloc(sizeof(u32) * (size_t)m * bw), *vC = xmalloc(sizeof(u32) * (size_t)bw * bw), *vR = xmalloc(sizeof(u32) * (size_t)m * bw);
        for (u32 i = 0; i < m; ++i) for (u32 j = 0; j < bw; ++j) blk[(size_t)i * bw + j] = A[(size_t)i * n + c0 + j];
        if (!sparse_alternate(vC, vR, blk, m, bw, maxnumcoeff)) rc = 1;                          /* :714 */
        for (u32 i = 0; i < m; ++i) for (u32 j = 0; j < bw; ++j) Res[(size_t)i * n + c0 + j] = vR[(size_t)i * bw + j];   /* augmentedMatrix :67-86 */
        if (reduced) {                                          /* CoB^T = [ L_blk . vC^T ... ] :726-740 */
            for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < bw; ++j) { u32 s = 0; for (u32 t = 0; t < bw; ++t) s = fadd(s, fmul(L[(size_t)i * n + c0 + t], vC[(size_t)j * bw + t])); TCoB[(size_t)i * n + c0 + j] = s; }
        } else for (u32 i = 0; i < bw; ++i) for (u32 j = 0; j < bw; ++j) CoB[(size_t)(c0 + i) * n + c0 + j] = vC[(size_t)i * bw + j];   /* diagonalMatrix :49-63 */
        free(blk); free(vC); free(vR);
    }
    if (reduced) transpose(CoB, TCoB, n, n);
    free(U); free(L); free(A); free(TCoB);
    if (candidates) *candidates = g_candidates;
    return rc;
}

/* how many candidates of the last plo_oracle_sparsify call carried a coordinate outside their block (tests: the fixture that
 * pins plinopt_sparsify.inl:305 must make this positive) */
uint64_t plo_oracle_sparsify_carried(void) { return g_carried; }
/* how many rows of the last call were filled by the canonical fallback (:317-326) */
uint64_t plo_oracle_sparsify_fallbacks(void) { return g_fallbacks; }
