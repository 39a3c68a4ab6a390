/* ==========================================================================
 * oracle/plo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of PLinOpt's per-candidate `Optimizer()` and of
 * the `CSEOptimiser` restart loop over a prime field Z_p (p < 2^31), used as
 * the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (plinopt_amd/, libplinopt_hip.so, bin/) never does.
 *
 * PARITY STATUS: "parity unpinned" against the genuine reference binary (it
 * cannot be built here: LinBox/Givaro absent) and the
 * reference's tests hold no golden op-count/text for this path.  What IS
 * pinned (tests/test_oracle_*.py): every emitted SLP evaluates to the input
 * matrix (the reference's own `slpcheck` criterion, Makefile:85-87,
 * bin/FDT.sh:58), reported counts equal the op-count of the emitted text under
 * `lineOperations` rules (plinopt_programs.inl:116-133), the hand-checked
 * Winograd tie set of SURVEY.md 8c, and best costs <= the stored-SLP bounds.
 *
 * Citations are relative to /root/reference/.
 * ========================================================================== */
#ifndef PLO_ORACLE_H
#define PLO_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-candidate random stream (the build's frozen definition; the reference
 * uses a time-seeded thread_local Givaro::GivRandom, plinopt_optimize.inl:263-265):
 *   state0 = 1 + splitmix64(seed) mod (2^31-2)
 *   next() : state = 950706376 * state mod (2^31-1); return state   (GivRandom LCG)
 *   tie pick = next() mod #ties                                                       */
uint32_t plo_oracle_rng_state0(uint64_t seed);
uint32_t plo_oracle_rng_next(uint32_t *state);

/* One candidate: include/plinopt_optimize.inl:616-631 (Optimizer), preceded by
 * the driver's `input2Temps` preamble (plinopt_library.inl:319-331) when
 * text != NULL.  Matrix is CSR, columns sorted per row, values in [1,p).
 * letters = {ouv, tev, rav, inv} e.g. "otri".  *text (if non-NULL) receives a
 * malloc'd NUL-terminated program; free with plo_oracle_free.
 * Returns 0, or <0 on error. */
int plo_oracle_optimizer(uint32_t m, uint32_t n, const uint32_t *rowptr,
                         const uint32_t *col, const uint32_t *val, uint32_t p,
                         uint64_t seed, const char letters[4],
                         uint32_t *adds, uint32_t *muls, char **text);

/* LUOptimiser restart body (plinopt_optimize.inl:1056-1100): Optimizer() on the first matrix, then on the
 * second with the generator running on; (adds, muls) summed.  text1/text2 optional (malloc'd). */
int plo_oracle_chain(uint32_t m1, uint32_t n1, const uint32_t *rp1, const uint32_t *c1, const uint32_t *v1,
                     uint32_t m2, uint32_t n2, const uint32_t *rp2, const uint32_t *c2, const uint32_t *v2,
                     uint32_t p, uint64_t seed, uint32_t *adds, uint32_t *muls, char **text1, char **text2);

/* Costs of many seeds (count only).  OpenMP-parallel over seeds when
 * nthreads > 1.  seeds == NULL means seed0, seed0+1, ... */
int plo_oracle_cost_many(uint32_t m, uint32_t n, const uint32_t *rowptr,
                         const uint32_t *col, const uint32_t *val, uint32_t p,
                         const uint64_t *seeds, uint64_t seed0, uint64_t nseeds,
                         uint32_t *adds, uint32_t *muls, int nthreads);

/* Restart loop: plinopt_optimize.inl:1193-1247 (CSEOptimiser) with the total
 * order (cmpOpCount key, seed): cost_mode 0 = sum then adds (default,
 * plinopt_optimize.h:61-63), 1 = adds then muls (OPTIMIZE_ADDITIONS :54-55),
 * 2 = sum only (OPTIMIZE_SUMS :57-58). */
int plo_oracle_cse_search(uint32_t m, uint32_t n, const uint32_t *rowptr,
                          const uint32_t *col, const uint32_t *val, uint32_t p,
                          uint64_t seed0, uint64_t nseeds, int cost_mode,
                          uint32_t *best_adds, uint32_t *best_muls,
                          uint64_t *best_seed, int nthreads);

/* First-step tie set (for the hand-checked Winograd fixture): writes up to cap
 * triples (a,b,r) of maximal frequency in map order; returns their number and
 * the frequency in *maxfrq. */
int plo_oracle_first_ties(uint32_t m, uint32_t n, const uint32_t *rowptr,
                          const uint32_t *col, const uint32_t *val, uint32_t p,
                          uint32_t *tri_abr, int cap, uint32_t *maxfrq);

/* One (block,row) enumeration of localSparsifier (plinopt_sparsify.inl:282-314) = |coeffs|^4 calls of testLinComb
 * (:167-197), rank by Gaussian elimination of a copy per candidate as in the reference.  TM n x m, Cand n x n,
 * dense row major, residues mod p.  index = ((i*C+j)*C+k)*C+l of the winner. */
int plo_oracle_cob_search(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock,
                          const uint32_t *coeffs, uint32_t C, uint32_t p, int32_t w0, int32_t w1,
                          int32_t *zeros_v, int32_t *zeros_w, uint64_t *index, uint32_t *found);

/* naiveOps: plinopt_library.inl:227-235 */
void plo_oracle_naive_ops(uint32_t m, const uint32_t *rowptr,
                          const uint32_t *val, uint32_t p,
                          uint32_t *adds, uint32_t *muls);

/* -E: the schedule space of RecSub (plinopt_optimize.inl:889-982) walked by index, see plo_oracle.c */
int plo_oracle_enum_optimizer(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                              uint64_t index, const char letters[4], uint32_t *adds, uint32_t *muls, uint64_t *prod, char **text);
int plo_oracle_enum_cost_many(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                              uint64_t first, uint64_t count, uint32_t *adds, uint32_t *muls, uint64_t *prods, int nthreads);

/* ---- trilplacer (plo_tril_oracle.c): in-place trilinear programs, plinopt_inplace.inl:400-502, :732-929.
 * A (m x nA), B (m x nB), T = transpose of the product matrix (m x nT): CSR, columns sorted per row, rational values
 * num/den (den == NULL: integers).  One candidate = one restart of SearchTriLinearAlgorithm :837-924 = row permutation
 * + coherent row negations from the seed's stream, then the oriented (variant 0) and the unoriented (variant 1) program;
 * ops6 = {ADD,SCA,MUL} of variant 0 then of variant 1.  seed == ~0 is the unpermuted oriented program of :829.
 * Returns 0, -1 bad dimensions, -3 rational overflow. */
#define PLO_TRIL_ARGS(x_) uint32_t n##x_, const uint32_t *rp##x_, const uint32_t *col##x_, const int64_t *num##x_, const int64_t *den##x_
int plo_oracle_tril_cost_many(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T),
                              const uint64_t *seeds, uint64_t seed0, uint64_t nseeds, uint32_t *ops6, int nthreads);
int plo_oracle_tril_program(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T),
                            uint64_t seed, int variant, uint32_t *ops6, char **text);
int plo_oracle_tril_search(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T),
                           uint64_t seed0, uint64_t nseeds, uint32_t *best_ops3, uint64_t *best_seed, uint32_t *best_variant);
/* The same with `expanded` (trilplacer -e, src/trilplacer.cpp:114-137): the c program is TransposedDoubleAlgorithm
 * (plinopt_inplace.inl:507-598) on DoubleExpand(T) (:676-716, built inside: 2m rows, nT+1 columns), every AXPY adds the
 * double-size product to two entries of c (MULTD, plinopt_inplace.h:111) and the AXPY count is halved (:799). */
int plo_oracle_tril_cost_many_x(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T), int expanded,
                                const uint64_t *seeds, uint64_t seed0, uint64_t nseeds, uint32_t *ops6);
int plo_oracle_tril_program_x(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T), int expanded,
                              uint64_t seed, int variant, uint32_t *ops6, char **text);
int plo_oracle_tril_search_x(uint32_t m, PLO_TRIL_ARGS(A), PLO_TRIL_ARGS(B), PLO_TRIL_ARGS(T), int expanded,
                             uint64_t seed0, uint64_t nseeds, uint32_t *best_ops3, uint64_t *best_seed, uint32_t *best_variant);

/* Literal RecSub / RecOptimizer (plinopt_optimize.inl:889-1013) with their savings accounting, for toy inputs (the tree is
 * exponential): *adds and *muls_recsub = the best (additions, multiplications) RecSub reports (:958-959 order; the
 * multiplications count every non +-1 entry left as one), *muls_final = the multiplications RecOptimizer returns after
 * ProgramGen (:1012), *nodes = RemOneCSE calls made. */
int plo_oracle_recsub(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                      uint32_t *adds, uint32_t *muls_recsub, uint32_t *muls_final, uint64_t *nodes);

/* One restart of KernelOptimiser (plinopt_optimize.inl:1299-1340) with the build's decomposition rule restated independently
 * (see plo_oracle.c): decomposition from the seed's stream, then Optimizer on Free and on Dep with one stream restarted
 * from the seed.  (adds, muls) summed; rank, NotIndep and the number of dependent rows computed by Dep are returned for the
 * log line.  0, or -2 for a zero dimensional kernel. */
int plo_oracle_kernel_restart(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p, uint64_t seed,
                              uint32_t *adds, uint32_t *muls, uint32_t *rank, uint32_t *notindep, uint32_t *ndep);

/* The LU factors bin/optimizer -G hands to the two chained Optimizer calls (LUOptimiser, plinopt_optimize.inl:1021-1109), with the build's
 * pivot rule (LinBox's is not in the reference tree): pivot row = smallest unused row holding an entry, pivot column = its smallest
 * column; restated on dense arrays, independently of host/plo_host.hpp `sparse_lu`.  U (m x n, row major, rows >= rank zero, columns
 * permuted: pivot columns first, then the others in increasing order) and L (m x m: multipliers, unit diagonal on the pivot rows; rows
 * permuted: pivot rows first) are written into caller arrays of m*n and m*m words; *rank is set.  Returns 0. */
int plo_oracle_lu(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                  uint32_t *U, uint32_t *L, uint32_t *rank);

/* The factorization M = Alt . CoB that bin/optimizer -A hands to the two chained Optimizer calls (ABOptimiser, plinopt_optimize.inl:1113-1187;
 * Factorizer / backSolver, plinopt_sparsify.inl:756-867, 924-984) with the build's rule, restated independently of host/plo_host.hpp
 * `ab_backsolve`: per back-solve (seed seed0 + i, i < loops) a random order of the rows; position i < n takes the first later row that
 * raises the rank of the rows before it (ranks recomputed from scratch); the first k rows of the resulting order are CoB (k x n) and unit
 * rows of Alt (m x k); every other row is row . B^-1 with B the first n rows (inverse by Gauss-Jordan).  Best = smallest
 * (nnz(Alt), non +-1 entries of Alt, nnz(CoB)), earlier seed on ties, starting from (M | 0) . (I ; 0).  Alt (m*k words) and CoB (k*n
 * words) row major, score[3].  Returns 0, or -2 when m <= n (identity / skipped). */
int plo_oracle_ab_factor(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                         uint64_t seed0, uint32_t loops, uint32_t k, uint32_t *Alt, uint32_t *CoB, uint32_t *score);

/* One decomposition of AllKernelOpt (-N, plinopt_optimize.inl:1357-1418): the rows in the PRESCRIBED order `ord`, NotIndep = first draw of the
 * stream of dseed, then the two Optimizer calls from the stream of cseed.  dep_out (m words) receives the kept dependent rows, depcols
 * (m*m bytes, zeroed by the caller) the column pattern of Dep.  0, or -2 for a zero dimensional kernel. */
int plo_oracle_kernel_order(uint32_t m, uint32_t n, const uint32_t *rowptr, const uint32_t *col, const uint32_t *val, uint32_t p,
                            const uint32_t *ord, uint64_t dseed, uint64_t cseed,
                            uint32_t *adds, uint32_t *muls, uint32_t *rank, uint32_t *notindep, uint32_t *ndep, uint32_t *dep_out, unsigned char *depcols);

void plo_oracle_free(void *ptr);
int plo_oracle_max_threads(void);

/* ---- bin/sparsifier as a whole (plo_sparsify_oracle.c; reference include/plinopt_sparsify.inl) */
/* the coefficient set of localSparsifier (:256-268, augment :21-35) for the n x m matrix TM (dense, row major) */
int plo_oracle_sp_coeffs(uint32_t n, uint32_t m, const uint32_t *TM, uint32_t p, uint32_t maxnumcoeff, uint32_t *out, uint32_t *ncoeffs);
/* one localSparsifier call (:206-347: nullspace seed, enumeration through testLinComb, canonical fallback): TM (n x m) and TCoB (n x n) are updated */
int plo_oracle_sp_local(uint32_t n, uint32_t m, uint32_t *TM, uint32_t *TCoB, uint32_t p, uint32_t maxnumcoeff);
/* blockSparsifier (:667-748) with sparseAlternate / SparseFactor / FactorDiagonals / sparseLU: M (m x n) -> CoB (n x n), Res (m x n), M == Res . CoB */
int plo_oracle_sparsify(uint32_t m, uint32_t n, const uint32_t *M, uint32_t p, uint32_t blocksize, uint32_t maxnumcoeff, int initial_elimination,
                        uint32_t *CoB, uint32_t *Res, uint64_t *candidates);
/* the same two over Q, the field the reference runs bin/sparsifier in without -q (src/sparsifier.cpp:66-83): the SAME restatement
 * (plo_sparsify_body.h) instantiated with checked 64-bit rationals; matrices as (numerators, denominators > 0), results in lowest terms */
int plo_oracle_sp_coeffs_q(uint32_t n, uint32_t m, const int64_t *TMnum, const int64_t *TMden, uint32_t maxnumcoeff, int64_t *outnum, int64_t *outden, uint32_t *ncoeffs);
int plo_oracle_sparsify_q(uint32_t m, uint32_t n, const int64_t *Mnum, const int64_t *Mden, uint32_t blocksize, uint32_t maxnumcoeff, int initial_elimination,
                          int64_t *CoBnum, int64_t *CoBden, int64_t *Resnum, int64_t *Resden, uint64_t *candidates);
/* counters of the last plo_oracle_sparsify[_q] call: candidates that carried a coordinate outside their block (:305), rows filled by the fallback (:317-326) */
uint64_t plo_oracle_sparsify_carried(void);
uint64_t plo_oracle_sparsify_fallbacks(void);

#ifdef __cplusplus
}
#endif
#endif
