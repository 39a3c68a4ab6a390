/* ===========================================================================
 * plo_sparsify_body.h -- the restatement of /root/reference/include/plinopt_sparsify.inl over an ABSTRACT field, included by
 * plo_sparsify_oracle.c once per field (Z_p: residues in uint32_t; Q: checked 64-bit rationals).  TEST INFRASTRUCTURE ONLY.
 *
 * The including file defines, before each inclusion:
 *   FN(name)            the name of this instance's function `name`
 *   elt                 the element type; ALL-ZERO BYTES MUST BE THE FIELD'S ZERO (calloc'ed arrays are zero matrices)
 *   F_ONE               the unit
 *   F_IS0(x) F_EQ(a,b)  tests
 *   F_LESS(a,b)         the order of std::map<Element,int> (FactorDiagonals :363-369): residues as integers, rationals by value
 *   F_ADD F_NEG F_MUL F_INV
 *   FN(build_coeffs)(elt *out, const elt *TM, u32 n, u32 m, u32 maxnumcoeff)    the coefficient list :256-268 (field specific:
 *                       over Modular<Integer> the list holds unreduced integers, over Q plain rationals)
 * Line references: see the header of plo_sparsify_oracle.c.
 * =========================================================================== */

/* rank of an r x c matrix by elimination of a COPY (rank :38-45) */
static u32 FN(rank_of)(const elt *A, u32 r, u32 c)
{
    elt *W = xmalloc(sizeof(elt) * (size_t)r * c); memcpy(W, A, sizeof(elt) * (size_t)r * c);
    u32 rk = 0;
    for (u32 col = 0; col < c && rk < r; ++col) {
        u32 p = rk; while (p < r && F_IS0(W[(size_t)p * c + col])) ++p;
        if (p == r) continue;
        if (p != rk) for (u32 j = 0; j < c; ++j) { elt t = W[(size_t)p * c + j]; W[(size_t)p * c + j] = W[(size_t)rk * c + j]; W[(size_t)rk * c + j] = t; }
        const elt iv = F_INV(W[(size_t)rk * c + col]);
        for (u32 i = rk + 1; i < r; ++i) if (!F_IS0(W[(size_t)i * c + col])) {
            const elt l = F_MUL(W[(size_t)i * c + col], iv);
            for (u32 j = col; j < c; ++j) W[(size_t)i * c + j] = F_ADD(W[(size_t)i * c + j], F_NEG(F_MUL(l, W[(size_t)rk * c + j])));
        }
        ++rk;
    }
    free(W);
    return rk;
}
static void FN(matmul)(elt *C, const elt *A, const elt *B, u32 r, u32 k, u32 c)     /* C (r x c) = A (r x k) . B (k x c) */
{
    for (u32 i = 0; i < r; ++i) for (u32 j = 0; j < c; ++j) {
        elt s; memset(&s, 0, sizeof s);
        for (u32 t = 0; t < k; ++t) s = F_ADD(s, F_MUL(A[(size_t)i * k + t], B[(size_t)t * c + j]));
        C[(size_t)i * c + j] = s;
    }
}
static void FN(transpose)(elt *T, const elt *A, u32 r, u32 c) { for (u32 i = 0; i < r; ++i) for (u32 j = 0; j < c; ++j) T[(size_t)j * r + i] = A[(size_t)i * c + j]; }
static u32 FN(density)(const elt *A, u32 r, u32 c) { u32 s = 0; for (size_t k = 0; k < (size_t)r * c; ++k) if (!F_IS0(A[k])) ++s; return s; }
/* inverse of an n x n matrix (Gauss-Jordan on [A | I]); 0 when singular */
static int FN(inverse)(elt *I, const elt *A, u32 n)
{
    elt *W = xmalloc(sizeof(elt) * (size_t)n * 2 * n);
    for (u32 i = 0; i < n; ++i) { for (u32 j = 0; j < n; ++j) W[(size_t)i * 2 * n + j] = A[(size_t)i * n + j]; W[(size_t)i * 2 * n + n + i] = F_ONE; }
    for (u32 col = 0; col < n; ++col) {
        u32 p = col; while (p < n && F_IS0(W[(size_t)p * 2 * n + col])) ++p;
        if (p == n) { free(W); return 0; }
        if (p != col) for (u32 j = 0; j < 2 * n; ++j) { elt t = W[(size_t)p * 2 * n + j]; W[(size_t)p * 2 * n + j] = W[(size_t)col * 2 * n + j]; W[(size_t)col * 2 * n + j] = t; }
        const elt iv = F_INV(W[(size_t)col * 2 * n + col]);
        for (u32 j = 0; j < 2 * n; ++j) W[(size_t)col * 2 * n + j] = F_MUL(W[(size_t)col * 2 * n + j], iv);
        for (u32 i = 0; i < n; ++i) if (i != col && !F_IS0(W[(size_t)i * 2 * n + col])) {
            const elt l = W[(size_t)i * 2 * n + col];
            for (u32 j = 0; j < 2 * n; ++j) W[(size_t)i * 2 * n + j] = F_ADD(W[(size_t)i * 2 * n + j], F_NEG(F_MUL(l, W[(size_t)col * 2 * n + j])));
        }
    }
    for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < n; ++j) I[(size_t)i * n + j] = W[(size_t)i * 2 * n + n + j];
    free(W);
    return 1;
}

/* ---- testLinComb :167-197 (Cand: n x n, a copy whose row `num` is replaced by w) */
static int FN(test_lin_comb)(int *w0, int *w1, elt *LCoB, elt *Cand, u32 num, const elt *w, const elt *TM, u32 n, u32 m)
{
    memcpy(Cand + (size_t)num * n, w, sizeof(elt) * n);         /* setRow(Cand,num,w) :172 */
    if (FN(rank_of)(Cand, n, n) > num) {                        /* :173-175 */
        int rl = 0, cl = 0;
        for (u32 c = 0; c < m; ++c) {                           /* applyTranspose, zeros :176-179 */
            elt s; memset(&s, 0, sizeof s);
            for (u32 i = 0; i < n; ++i) s = F_ADD(s, F_MUL(w[i], TM[(size_t)i * m + c]));
            if (F_IS0(s)) ++rl;
        }
        for (u32 i = 0; i < n; ++i) if (F_IS0(w[i])) ++cl;      /* :180 */
        if (rl > *w0 || (rl == *w0 && cl > *w1)) { *w0 = rl; *w1 = cl; memcpy(LCoB + (size_t)num * n, w, sizeof(elt) * n); return 1; }   /* :183-194 */
    }
    return 0;
}

/* ---- localSparsifier :206-347; TM: n x m, TCoB: n x n, both updated */
static void FN(local_sparsifier)(elt *TCoB, elt *TM, u32 n, u32 m, u32 maxnumcoeff)
{
    elt *LCoB = xmalloc(sizeof(elt) * (size_t)n * n);
    int cnHw = -1, rnHw = -1;
    if (n > 1) {                                                /* :227-252 */
        /* N = TM^T (m x n), rows sorted by decreasing size, stable; rows dropped from the end while the rank is n */
        elt *N = xmalloc(sizeof(elt) * (size_t)m * n); u32 *ord = xmalloc(sizeof(u32) * m), *sz = xmalloc(sizeof(u32) * m);
        for (u32 c = 0; c < m; ++c) { ord[c] = c; sz[c] = 0; for (u32 i = 0; i < n; ++i) if (!F_IS0(TM[(size_t)i * m + c])) ++sz[c]; }
        for (u32 a = 1; a < m; ++a) { const u32 x = ord[a]; u32 b = a; while (b > 0 && sz[ord[b - 1]] < sz[x]) { ord[b] = ord[b - 1]; --b; } ord[b] = x; }   /* insertion sort: stable */
        for (u32 k = 0; k < m; ++k) for (u32 i = 0; i < n; ++i) N[(size_t)k * n + i] = TM[(size_t)i * m + ord[k]];
        u32 rows = m;
        while (rows > 0 && FN(rank_of)(N, rows, n) == n) --rows;    /* :231-233 */
        if (rows > 0) {
            /* reduced row echelon form of the kept rows; the vector of the FIRST free column, free variable = 1 (the build's rule) */
            elt *W = xmalloc(sizeof(elt) * (size_t)rows * n); memcpy(W, N, sizeof(elt) * (size_t)rows * n);
            u32 *pivcol = xmalloc(sizeof(u32) * n); u32 rk = 0;
            for (u32 col = 0; col < n && rk < rows; ++col) {
                u32 p = rk; while (p < rows && F_IS0(W[(size_t)p * n + col])) ++p;
                if (p == rows) continue;
                if (p != rk) for (u32 j = 0; j < n; ++j) { elt t = W[(size_t)p * n + j]; W[(size_t)p * n + j] = W[(size_t)rk * n + j]; W[(size_t)rk * n + j] = t; }
                const elt iv = F_INV(W[(size_t)rk * n + col]);
                for (u32 j = 0; j < n; ++j) W[(size_t)rk * n + j] = F_MUL(W[(size_t)rk * n + j], iv);
                for (u32 i = 0; i < rows; ++i) if (i != rk && !F_IS0(W[(size_t)i * n + col])) {
                    const elt l = W[(size_t)i * n + col];
                    for (u32 j = 0; j < n; ++j) W[(size_t)i * n + j] = F_ADD(W[(size_t)i * n + j], F_NEG(F_MUL(l, W[(size_t)rk * n + j])));
                }
                pivcol[rk++] = col;
            }
            u32 fc = n;
            for (u32 col = 0; col < n && fc == n; ++col) { int isp = 0; for (u32 k = 0; k < rk; ++k) if (pivcol[k] == col) isp = 1; if (!isp) fc = col; }
            if (fc < n) {
                memset(LCoB, 0, sizeof(elt) * n);
                LCoB[fc] = F_ONE;
                for (u32 k = 0; k < rk; ++k) LCoB[pivcol[k]] = F_NEG(W[(size_t)k * n + fc]);            /* x_pivot = - entry of the free column */
                cnHw = 0; for (u32 i = 0; i < n; ++i) if (!F_IS0(LCoB[i])) ++cnHw;                       /* LCoB[0].size() :245 */
                rnHw = 0;
                for (u32 c = 0; c < m; ++c) {                                                           /* :243-246 */
                    elt s; memset(&s, 0, sizeof s);
                    for (u32 i = 0; i < n; ++i) s = F_ADD(s, F_MUL(LCoB[i], TM[(size_t)i * m + c]));
                    if (F_IS0(s)) ++rnHw;
                }
            }
            free(W); free(pivcol);
        }
        free(N); free(ord); free(sz);
    }
    elt *Coeffs = xmalloc(sizeof(elt) * (maxnumcoeff + 8));
    const u32 C = FN(build_coeffs)(Coeffs, TM, n, m, maxnumcoeff);
    const u32 numlargeblocks = n >> 2, lastblock = n - (numlargeblocks << 2), numblocks = lastblock ? numlargeblocks + 1 : numlargeblocks;   /* :274-277 */
    const u32 multiple = numblocks << 2;                        /* :277 (>= n) */
    elt *A = xmalloc(sizeof(elt) * (size_t)n * n), *w = xmalloc(sizeof(elt) * ((size_t)multiple + 4));
    for (u32 block = 0; block < numblocks; ++block) {
        memset(w, 0, sizeof(elt) * ((size_t)multiple + 4));     /* w.resize(0); w.resize(multiple) :283 -- ONCE PER BLOCK */
        const u32 off = block << 2, first = n - off < 4 ? n - off : 4;
        for (u32 num = 0; num < first; ++num) {
            memcpy(A, LCoB, sizeof(elt) * (size_t)n * n);       /* matrixCopy(A, LCoB) :288 */
            int w0 = -1, w1 = -1; int found = (block == 0 && num == 0);
            if (found) { w0 = rnHw; w1 = cnHw; }                /* :289-294 */
            for (u32 i = 0; i < C; ++i) for (u32 j = 0; j < C; ++j) for (u32 k = 0; k < C; ++k) for (u32 l = 0; l < C; ++l) {   /* :299-314 */
                /* w.resize(multiple) :305 zero-fills the positions n.. only: what w holds below n STAYS -- after the canonical
                 * fallback of an earlier row of this block (below) that is its w[p] = 1, for p outside the block */
                memset(w + n, 0, sizeof(elt) * ((size_t)multiple + 4 - n));
                w[off] = Coeffs[i]; w[off + 1] = Coeffs[j]; w[off + 2] = Coeffs[k]; w[off + 3] = Coeffs[l];      /* :306-309 */
                /* w.resize(TM.rowdim()) :311: positions beyond n are dropped (test_lin_comb reads n words) */
                ++g_candidates;
                for (u32 x = 0; x < n; ++x) if (!F_IS0(w[x]) && (x < off || x >= off + 4)) { ++g_carried; break; }
                found |= FN(test_lin_comb)(&w0, &w1, LCoB, A, num + off, w, TM, n, m);
            }
            for (u32 pp = 0; !found && pp < n; ++pp) {           /* canonical fallback :317-326 */
                w0 = -1; w1 = -1;
                memset(w, 0, sizeof(elt) * ((size_t)multiple + 4)); w[pp] = F_ONE;      /* w.resize(0); w.resize(rowdim); w[p]=1 :320-321 */
                if (pp == 0) ++g_fallbacks;
                found |= FN(test_lin_comb)(&w0, &w1, LCoB, A, num + off, w, TM, n, m);
            }
        }
    }
    elt *TR = xmalloc(sizeof(elt) * (size_t)n * m), *TS = xmalloc(sizeof(elt) * (size_t)n * n);
    FN(matmul)(TR, LCoB, TM, n, n, m); FN(matmul)(TS, LCoB, TCoB, n, n, n);                /* :336-344 */
    memcpy(TM, TR, sizeof(elt) * (size_t)n * m); memcpy(TCoB, TS, sizeof(elt) * (size_t)n * n);
    free(LCoB); free(Coeffs); free(A); free(w); free(TR); free(TS);
}

/* ---- FactorDiagonals :355-375 (std::map by value, max_element = first of the largest counts) */
static void FN(factor_diagonals)(elt *TCoB, elt *TM, u32 n, u32 m)
{
    for (u32 i = 0; i < n; ++i) {
        elt best; memset(&best, 0, sizeof best); int bestc = 0;
        for (u32 j = 0; j < m; ++j) { const elt v = TM[(size_t)i * m + j]; if (F_IS0(v)) continue;
            int c = 0; for (u32 t = 0; t < m; ++t) if (F_EQ(TM[(size_t)i * m + t], v)) ++c;
            if (c > bestc || (c == bestc && F_LESS(v, best))) { bestc = c; best = v; } }         /* ascending keys, strict > keeps the smallest */
        if (!bestc || F_EQ(best, F_ONE)) continue;
        const elt ir = F_INV(best);
        for (u32 j = 0; j < m; ++j) TM[(size_t)i * m + j] = F_MUL(TM[(size_t)i * m + j], ir);
        for (u32 j = 0; j < n; ++j) TCoB[(size_t)i * n + j] = F_MUL(TCoB[(size_t)i * n + j], ir);
    }
}

/* ---- SparseFactor :474-513 */
static u32 FN(sparse_factor)(elt *TICoB, elt *TM, u32 n, u32 m, u32 start, u32 increment, u32 threshold)
{
    u32 s2 = FN(density)(TM, n, m), ss, numcoeffs = start;
    do {
        ss = s2;
        FN(local_sparsifier)(TICoB, TM, n, m, numcoeffs);
        FN(factor_diagonals)(TICoB, TM, n, m);
        s2 = FN(density)(TM, n, m);
        if (numcoeffs < threshold) numcoeffs += increment;
    } while (s2 < ss);
    return s2;
}

/* ---- sparseLU :524-566 with the build's pivot rule: A (r x c) <- U.P, QL (r x r) <- Q.L, only when sparser */
static int FN(sparse_lu)(elt *QL, elt *A, u32 r, u32 c, u32 sparsity)
{
    elt *W = xmalloc(sizeof(elt) * (size_t)r * c), *Lm = xmalloc(sizeof(elt) * (size_t)r * r);   /* Lm[i][k]: multiplier of row i on pivot k */
    memcpy(W, A, sizeof(elt) * (size_t)r * c);
    int *isp = xmalloc(sizeof(int) * r); u32 *prow = xmalloc(sizeof(u32) * r); u32 rk = 0;
    for (;;) {
        u32 pr = r, pc = 0;
        for (u32 i = 0; i < r && pr == r; ++i) { if (isp[i]) continue; for (u32 j = 0; j < c; ++j) if (!F_IS0(W[(size_t)i * c + j])) { pr = i; pc = j; break; } }
        if (pr == r) break;
        isp[pr] = 1; prow[rk] = pr;
        const elt iv = F_INV(W[(size_t)pr * c + pc]);
        for (u32 i = 0; i < r; ++i) {
            if (isp[i] || F_IS0(W[(size_t)i * c + pc])) continue;
            const elt l = F_MUL(W[(size_t)i * c + pc], iv);
            for (u32 j = 0; j < c; ++j) W[(size_t)i * c + j] = F_ADD(W[(size_t)i * c + j], F_NEG(F_MUL(l, W[(size_t)pr * c + j])));
            Lm[(size_t)i * r + rk] = l;
        }
        ++rk;
    }
    u32 dens = 0; for (u32 k = 0; k < rk; ++k) for (u32 j = 0; j < c; ++j) if (!F_IS0(W[(size_t)prow[k] * c + j])) ++dens;
    int sparser = dens < sparsity;                              /* density(U) < sparsity :551 */
    if (sparser) {
        /* A <- the pivot rows in pivot order (zero rows behind); QL[i][k] = 1 for pivot k's own row, the multiplier for a reduced row;
           a row that never was a pivot gets a unit in its own column behind the rank (those rows of the new A are zero) */
        elt *nA = xmalloc(sizeof(elt) * (size_t)r * c), *nQ = xmalloc(sizeof(elt) * (size_t)r * r);
        for (u32 k = 0; k < rk; ++k) { memcpy(nA + (size_t)k * c, W + (size_t)prow[k] * c, sizeof(elt) * c); nQ[(size_t)prow[k] * r + k] = F_ONE; }
        u32 nx = rk;
        for (u32 i = 0; i < r; ++i) {
            for (u32 k = 0; k < rk; ++k) if (!F_IS0(Lm[(size_t)i * r + k])) nQ[(size_t)i * r + k] = Lm[(size_t)i * r + k];
            if (!isp[i]) nQ[(size_t)i * r + nx++] = F_ONE;
        }
        memcpy(A, nA, sizeof(elt) * (size_t)r * c); memcpy(QL, nQ, sizeof(elt) * (size_t)r * r);
        free(nA); free(nQ);
    }
    free(W); free(Lm); free(isp); free(prow);
    return sparser;
}
/* ---- sparseILU :574-600 */
static int FN(sparse_ilu)(elt *TC, elt *A, u32 r, u32 c, u32 sparsity)
{
    elt *QL = xmalloc(sizeof(elt) * (size_t)r * r); for (u32 i = 0; i < r; ++i) QL[(size_t)i * r + i] = F_ONE;
    const int sparser = FN(sparse_lu)(QL, A, r, c, sparsity);
    if (sparser) {
        elt *I = xmalloc(sizeof(elt) * (size_t)r * r), *K = xmalloc(sizeof(elt) * (size_t)r * r);
        if (!FN(inverse)(I, QL, r)) abort();
        FN(matmul)(K, I, TC, r, r, r);                          /* applyInverse: TC == QL . K */
        memcpy(TC, K, sizeof(elt) * (size_t)r * r);
        free(I); free(K);
    }
    free(QL);
    return sparser;
}

/* ---- sparseAlternate :610-661: M (m x n) -> CoB (n x n), Res (m x n) */
static int FN(sparse_alternate)(elt *CoB, elt *Res, const elt *M, u32 m, u32 n, u32 maxnumcoeff)
{
    elt *TM = xmalloc(sizeof(elt) * (size_t)n * m), *TICoB = xmalloc(sizeof(elt) * (size_t)n * n);
    FN(transpose)(TM, M, m, n);
    for (u32 i = 0; i < n; ++i) TICoB[(size_t)i * n + i] = F_ONE;
    FN(factor_diagonals)(TICoB, TM, n, m);                      /* :627 */
    FN(sparse_ilu)(TICoB, TM, n, m, FN(density)(TM, n, m));     /* :629 */
    FN(sparse_factor)(TICoB, TM, n, m, 3, 4, 11);               /* defaults, plinopt_sparsify.h:78-80 */
    FN(sparse_factor)(TICoB, TM, n, m, maxnumcoeff, 1, maxnumcoeff);   /* :641 */
    elt *I = xmalloc(sizeof(elt) * (size_t)n * n);
    const int ok = FN(inverse)(I, TICoB, n);
    if (ok) { FN(transpose)(CoB, I, n, n); FN(transpose)(Res, TM, n, m); }   /* inverseTranspose :646, Transpose :651 */
    free(TM); free(TICoB); free(I);
    return ok;
}

/* blockSparsifier :667-748: M (m x n, dense, row major) -> CoB (n x n), Res (m x n) with M == Res . CoB; returns 0, 1 = a singular change of basis */
static int FN(block_sparsifier)(u32 m, u32 n, const elt *M, u32 blocksize, u32 maxnumcoeff, int initial_elimination, elt *CoB, elt *Res)
{
    int rc = 0;
    if (blocksize <= 1) return FN(sparse_alternate)(CoB, Res, M, m, n, maxnumcoeff) ? 0 : 1;
    elt *U = xmalloc(sizeof(elt) * (size_t)n * m), *L = xmalloc(sizeof(elt) * (size_t)n * n);
    int reduced = initial_elimination;
    if (initial_elimination) {
        FN(transpose)(U, M, m, n);
        for (u32 i = 0; i < n; ++i) L[(size_t)i * n + i] = F_ONE;
        reduced = FN(sparse_lu)(L, U, n, m, FN(density)(U, n, m));      /* :688 */
    }
    elt *A = xmalloc(sizeof(elt) * (size_t)m * n);
    if (reduced) FN(transpose)(A, U, n, m); else memcpy(A, M, sizeof(elt) * (size_t)m * n);        /* :699 */
    memset(Res, 0, sizeof(elt) * (size_t)m * n); memset(CoB, 0, sizeof(elt) * (size_t)n * n);
    elt *TCoB = xmalloc(sizeof(elt) * (size_t)n * n);
    for (u32 c0 = 0; c0 < n; c0 += blocksize) {                 /* separateColumnBlocks :89-117 */
        const u32 bw = n - c0 < blocksize ? n - c0 : blocksize;
        elt *blk = xmalloc(sizeof(elt) * (size_t)m * bw), *vC = xmalloc(sizeof(elt) * (size_t)bw * bw), *vR = xmalloc(sizeof(elt) * (size_t)m * bw);
        for (u32 i = 0; i < m; ++i) for (u32 j = 0; j < bw; ++j) blk[(size_t)i * bw + j] = A[(size_t)i * n + c0 + j];
        if (!FN(sparse_alternate)(vC, vR, blk, m, bw, maxnumcoeff)) rc = 1;                      /* :714 */
        for (u32 i = 0; i < m; ++i) for (u32 j = 0; j < bw; ++j) Res[(size_t)i * n + c0 + j] = vR[(size_t)i * bw + j];   /* augmentedMatrix :67-86 */
        if (reduced) {                                          /* CoB^T = [ L_blk . vC^T ... ] :726-740 */
            for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < bw; ++j) {
                elt s; memset(&s, 0, sizeof s);
                for (u32 t = 0; t < bw; ++t) s = F_ADD(s, F_MUL(L[(size_t)i * n + c0 + t], vC[(size_t)j * bw + t]));
                TCoB[(size_t)i * n + c0 + j] = s;
            }
        } else for (u32 i = 0; i < bw; ++i) for (u32 j = 0; j < bw; ++j) CoB[(size_t)(c0 + i) * n + c0 + j] = vC[(size_t)i * bw + j];   /* diagonalMatrix :49-63 */
        free(blk); free(vC); free(vR);
    }
    if (reduced) FN(transpose)(CoB, TCoB, n, n);
    free(U); free(L); free(A); free(TCoB);
    return rc;
}
