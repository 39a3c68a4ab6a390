/* ===========================================================================
 * plo_sparsify_oracle.c -- CPU restatement of the reference's change-of-basis
 * search (bin/sparsifier), whole tool, over Z_p AND over Q (the field the reference
 * runs it in without -q, src/sparsifier.cpp:66-83).  TEST INFRASTRUCTURE ONLY: tests/
 * and the cpu_baseline leg of bench.py may call it, never the product.
 * ONE restatement (plo_sparsify_body.h, written against an abstract field), two instances.
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
 * Dense row-major arrays of field elements, plain loops, no sharing of code with plinopt_amd/.
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
 * Over Q (Givaro::QField<Rational>) elements are rationals in lowest terms with a positive denominator; here: 64-bit numerator
 * and denominator, every operation through 128-bit intermediates, abort() on a result that does not fit (never on the data set).
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

static uint64_t g_candidates, g_carried, g_fallbacks;      /* g_carried: candidates evaluated with a coordinate outside their block (the fallback's w[p] = 1 kept by :305) */

/* ======================================================================================================== instance 1: Z_p */
#define FN(name) zp_##name
#define elt u32
#define F_ONE (1 % P_)
#define F_IS0(x) ((x) == 0)
#define F_EQ(a, b) ((a) == (b))
#define F_LESS(a, b) ((a) < (b))
#define F_ADD(a, b) fadd(a, b)
#define F_NEG(a) fneg(a)
#define F_MUL(a, b) fmul(a, b)
#define F_INV(a) finv(a)

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
static u32 zp_build_coeffs(u32 *out, const u32 *TM, u32 n, u32 m, u32 maxnumcoeff)
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

#include "plo_sparsify_body.h"
#undef FN
#undef elt
#undef F_ONE
#undef F_IS0
#undef F_EQ
#undef F_LESS
#undef F_ADD
#undef F_NEG
#undef F_MUL
#undef F_INV

/* ======================================================================================================== instance 2: Q */
typedef struct { int64_t n; uint64_t dm1; } rat;               /* n / (dm1 + 1), lowest terms: all-zero bytes are 0/1 */
typedef __int128 i128;
static i128 gcd128(i128 a, i128 b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { i128 t = a % b; a = b; b = t; } return a; }
static rat rmake(i128 n, i128 d)
{
    if (d == 0) abort();
    if (d < 0) { n = -n; d = -d; }
    const i128 g = gcd128(n, d); if (g > 1) { n /= g; d /= g; }
    if (n > INT64_MAX || n < -INT64_MAX || d > INT64_MAX) abort();          /* outside the checker's range */
    rat r; r.n = (int64_t)n; r.dm1 = (uint64_t)(d - 1); return r;
}
#define RD(x) ((i128)(x).dm1 + 1)
static rat radd(rat a, rat b) { return rmake((i128)a.n * RD(b) + (i128)b.n * RD(a), RD(a) * RD(b)); }
static rat rneg(rat a) { a.n = -a.n; return a; }
static rat rmul(rat a, rat b) { return rmake((i128)a.n * b.n, RD(a) * RD(b)); }
static rat rinv(rat a) { return rmake(RD(a), (i128)a.n); }
static int rless(rat a, rat b) { return (i128)a.n * RD(b) < (i128)b.n * RD(a); }
static const rat R_ONE = {1, 0};
#define FN(name) q_##name
#define elt rat
#define F_ONE R_ONE
#define F_IS0(x) ((x).n == 0)
#define F_EQ(a, b) ((a).n == (b).n && (a).dm1 == (b).dm1)
#define F_LESS(a, b) rless(a, b)
#define F_ADD(a, b) radd(a, b)
#define F_NEG(a) rneg(a)
#define F_MUL(a, b) rmul(a, b)
#define F_INV(a) rinv(a)
/* ---- coefficient set :256-268 with augment :21-35 over QField<Rational>: {0, 1, -1}, then r, -r, 1/r, -1/r for every stored entry
 * r not yet listed (row major), then for 2, 3, ... until the list holds maxnumcoeff elements; cut to maxnumcoeff */
static u32 q_build_coeffs(rat *out, const rat *TM, u32 n, u32 m, u32 maxnumcoeff)
{
    u32 cn = 0, cap = maxnumcoeff + 8 + 4 * n * m;
    rat *c = xmalloc(sizeof(rat) * cap);
#define Q_AUGMENT(r_) do { const rat r__ = (r_); int f__ = 0; for (u32 k = 0; k < cn; ++k) if (F_EQ(c[k], r__)) f__ = 1; \
        if (!f__) { c[cn++] = r__; c[cn++] = rneg(r__); c[cn++] = rinv(r__); c[cn++] = rneg(rinv(r__)); } } while (0)
    c[cn++] = rmake(0, 1); c[cn++] = rmake(1, 1); c[cn++] = rmake(-1, 1);                        /* :256 */
    for (u32 i = 0; i < n; ++i) for (u32 j = 0; j < m; ++j) if (!F_IS0(TM[(size_t)i * m + j])) Q_AUGMENT(TM[(size_t)i * m + j]);   /* :257-261 */
    for (int64_t i = 2; cn < maxnumcoeff; ++i) Q_AUGMENT(rmake(i, 1));                            /* :263-265 */
#undef Q_AUGMENT
    if (cn > maxnumcoeff) cn = maxnumcoeff;                     /* :268 */
    memcpy(out, c, sizeof(rat) * cn);
    free(c);
    return cn;
}
#include "plo_sparsify_body.h"

/* ======================================================================================================== entry points */
/* the coefficient set alone (S1) */
int plo_oracle_sp_coeffs(uint32_t n, uint32_t m, const uint32_t *TM, uint32_t p, uint32_t maxnumcoeff, uint32_t *out, uint32_t *ncoeffs)
{
    P_ = p; *ncoeffs = zp_build_coeffs(out, TM, n, m, maxnumcoeff); return 0;
}
/* one localSparsifier call (S3, with seed vector and fallback): TM n x m and TCoB n x n are updated */
int plo_oracle_sp_local(uint32_t n, uint32_t m, uint32_t *TM, uint32_t *TCoB, uint32_t p, uint32_t maxnumcoeff)
{
    P_ = p; zp_local_sparsifier(TCoB, TM, n, m, maxnumcoeff); return 0;
}
/* blockSparsifier :667-748 over Z_p: M (m x n, dense, row major) -> CoB (n x n), Res (m x n) with M == Res . CoB; returns 0, 1 = a singular change of basis */
int plo_oracle_sparsify(uint32_t m, uint32_t n, const uint32_t *M, uint32_t p, uint32_t blocksize, uint32_t maxnumcoeff, int initial_elimination,
                        uint32_t *CoB, uint32_t *Res, uint64_t *candidates)
{
    P_ = p; g_candidates = 0; g_carried = 0; g_fallbacks = 0;
    const int rc = zp_block_sparsifier(m, n, M, blocksize, maxnumcoeff, initial_elimination, CoB, Res);
    if (candidates) *candidates = g_candidates;
    return rc;
}

/* ---- the same over Q: matrices as pairs of int64 arrays (numerators, denominators > 0) */
static rat *q_in(const int64_t *num, const int64_t *den, size_t k) { rat *a = xmalloc(sizeof(rat) * k); for (size_t i = 0; i < k; ++i) a[i] = rmake(num[i], den[i]); return a; }
static void q_out(int64_t *num, int64_t *den, const rat *a, size_t k) { for (size_t i = 0; i < k; ++i) { num[i] = a[i].n; den[i] = (int64_t)(a[i].dm1 + 1); } }
int plo_oracle_sp_coeffs_q(uint32_t n, uint32_t m, const int64_t *TMnum, const int64_t *TMden, uint32_t maxnumcoeff, int64_t *outnum, int64_t *outden, uint32_t *ncoeffs)
{
    rat *TM = q_in(TMnum, TMden, (size_t)n * m), *out = xmalloc(sizeof(rat) * (maxnumcoeff + 8));
    *ncoeffs = q_build_coeffs(out, TM, n, m, maxnumcoeff);
    q_out(outnum, outden, out, *ncoeffs);
    free(TM); free(out);
    return 0;
}
int plo_oracle_sparsify_q(uint32_t m, uint32_t n, const int64_t *Mnum, const int64_t *Mden, uint32_t blocksize, uint32_t maxnumcoeff, int initial_elimination,
                          int64_t *CoBnum, int64_t *CoBden, int64_t *Resnum, int64_t *Resden, uint64_t *candidates)
{
    g_candidates = 0; g_carried = 0; g_fallbacks = 0;
    rat *M = q_in(Mnum, Mden, (size_t)m * n), *CoB = xmalloc(sizeof(rat) * (size_t)n * n), *Res = xmalloc(sizeof(rat) * (size_t)m * n);
    const int rc = q_block_sparsifier(m, n, M, blocksize, maxnumcoeff, initial_elimination, CoB, Res);
    q_out(CoBnum, CoBden, CoB, (size_t)n * n); q_out(Resnum, Resden, Res, (size_t)m * n);
    free(M); free(CoB); free(Res);
    if (candidates) *candidates = g_candidates;
    return rc;
}

/* how many candidates of the last plo_oracle_sparsify[_q] call carried a coordinate outside their block (tests: the fixture that
 * pins plinopt_sparsify.inl:305 must make this positive) */
uint64_t plo_oracle_sparsify_carried(void) { return g_carried; }
/* how many rows of the last call were filled by the canonical fallback (:317-326) */
uint64_t plo_oracle_sparsify_fallbacks(void) { return g_fallbacks; }
