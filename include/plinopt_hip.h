/* ==========================================================================
 * plinopt_hip.h -- C-ABI of libplinopt_hip.so, the MI355X (gfx950) drop-in for
 * PLinOpt's randomized multi-start search for minimum-cost straight-line
 * programs.  The reference has no FFI: the seam is the body of its OpenMP
 * restart loops.  Each entry point names the reference code it replaces
 * (paths relative to the reference tree).  Plain pointers and sizes only.
 *
 * Randomness: the reference draws from a time-seeded thread_local
 * Givaro::GivRandom (include/plinopt_optimize.inl:263-265), so it has no
 * reproducible "seed".  Here every candidate owns a stream defined by its
 * 64-bit seed:  state0 = 1 + splitmix64(seed) mod (2^31-2);
 * next(): state = 950706376*state mod (2^31-1) (the GivRandom LCG);
 * tie pick = next() mod #ties, ties in std::map order of (col_a,col_b,ratio).
 * ========================================================================== */
#ifndef PLINOPT_HIP_H
#define PLINOPT_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* status codes (0 = ok; negative = error; text via plo_last_error()) */
#define PLO_OK            0
#define PLO_E_ARG        -1   /* bad argument (null pointer, p<2, unsorted row ...) */
#define PLO_E_HIP        -2   /* HIP runtime error / no device / extension missing  */
#define PLO_E_CAPACITY   -3   /* matrix too large for the LDS-resident kernel        */
#define PLO_E_UNSUPPORTED -4  /* feature not available on the device path            */
#define PLO_E_INTERNAL   -5   /* device-side consistency check failed                */

/* Sparse matrix over Z_p in CSR: what FMatrix lM(M,F) holds at
 * include/plinopt_optimize.inl:1206 (LinBox SparseMatrix<Field,SparseSeq> after
 * rebind: rows of (column, residue) sorted by column, residues in [1,p)). */
typedef struct {
    uint32_t m, n;            /* rows, columns */
    const uint32_t *rowptr;   /* m+1 */
    const uint32_t *col;      /* nnz, strictly increasing inside a row */
    const uint32_t *val;      /* nnz, canonical residues in [1,p) */
} plo_csr_t;

/* Best candidate: Pair<size_t> nbops of CSEOptimiser (adds, muls) + its seed.
 * adds = muls = UINT32_MAX is the reference's Pair<size_t>{-1,-1} "+infinity"
 * (include/plinopt_optimize.inl:883). */
typedef struct {
    uint32_t adds, muls;
    uint64_t seed;
} plo_best_t;

typedef struct {
    double   seconds;        /* host wall clock of the call (upload excluded if plan reused) */
    double   kernel_ms;      /* sum of search-kernel durations, HIP events on the launch stream */
    uint64_t candidates;     /* candidates evaluated */
    uint32_t launches;       /* kernel launches */
    uint32_t lds_bytes;      /* dynamic LDS per workgroup */
    uint32_t waves_per_wg;   /* candidates in flight per workgroup */
    uint32_t grid;           /* workgroups per launch */
    uint64_t algo_bytes;     /* algorithmic bytes per candidate, B_cand = 8*nnz + 12*P0 + 8 */
    uint32_t reduce;         /* plo_*_search_multi: 1 = the minimum over the devices is the result of the RCCL MIN all-reduce (and equals the host's) */
    uint32_t reserved;
    double   reduce_seconds; /* plo_*_search_multi: wall time inside the all-reduce (two 8-byte MIN all-reduces; communicator set-up excluded: cached) */
} plo_stats_t;

/* cost order of the restart loop, include/plinopt_optimize.h:53-64 */
#define PLO_COST_SUM_THEN_ADD 0   /* default cmpOpCount            */
#define PLO_COST_ADD_THEN_MUL 1   /* -DOPTIMIZE_ADDITIONS           */
#define PLO_COST_SUM          2   /* -DOPTIMIZE_SUMS                */
#define PLO_COST_RECSUB       3   /* schedule enumeration only: RecSub's order (:958-959) on RecSub's own counts (:950-951):
                                   * additions, then the multiplications BEFORE ProgramGen (multipliers emitted + non +-1
                                   * entries left); plo_best_t.muls then holds that count */

typedef struct plo_plan plo_plan_t;   /* a matrix prepared and resident in HBM */

/* Library life cycle.  device = HIP ordinal (one process per GPU). */
int plo_init(int device);
int plo_shutdown(void);
const char *plo_last_error(void);
int plo_device_count(void);

/* Upload one matrix (replaces the per-restart copies `FMatrix lM(M,F), lT(T,F)`
 * of include/plinopt_optimize.inl:1206-1207: the matrix is converted once). */
int plo_cse_plan_create(const plo_csr_t *A, uint32_t p, plo_plan_t **plan);
/* flags: PLO_PLAN_HBM forces the HBM-resident kernel family (one workgroup per candidate,
 * plo_cse_big.hip) that plo_cse_plan_create selects by itself when a candidate does not fit LDS.
 * Limits of that family (PLO_E_CAPACITY): 32766 rows, 32768 columns (created ones included), rows of 8192 entries, 65536 distinct
 * coefficients, any odd prime below 2^31 -- but a modulus too wide for a residue in the 48-bit pair key beside the columns (above
 * 2^(48 - 2 ceil(log2 columns)); a 31-bit prime: more than 256 columns) needs a matrix of at most 32 distinct coefficients (the key
 * then holds the ratio's identifier). */
#define PLO_PLAN_HBM 1u
int plo_cse_plan_create_ex(const plo_csr_t *A, uint32_t p, uint32_t flags, plo_plan_t **plan);
/* 1 if the plan runs on the HBM-resident kernel family, 0 for the LDS-resident wave kernel */
int plo_cse_plan_is_hbm(const plo_plan_t *plan);
/* Diagnostics of the HBM-resident family, summed over the candidates of the plan's last launch:
 * out[0] CSE steps (OneSub iterations, include/plinopt_optimize.inl:237-312), [1] full pair-table scans,
 * [2] frequency-level rebuilds, [3] tie picks (:260-265) resolved by bisection on the key (more ties in one
 * column than the LDS list holds), [4] pairs that went through the spill list (no room in the LDS aggregation
 * table), [5] sweeps whose claimed-slot list overflowed, [6] candidates, [7] how often the plan was rebuilt with the eager
 * pair table because a candidate outgrew the structures of the deferred updates (sized from the input's triples). */
int plo_cse_plan_hbm_counters(const plo_plan_t *plan, uint32_t out[8]);
/* the same, the first n <= 10 counters: [8] windows of the flat sweep beyond the first one of a batch of rows (a wave takes the entries
 * of its rows 2048 at a time; PLO_BIG_FWIN makes the window smaller in the tests), counted on wave 0; [9] rows walked by the row search
 * (the length of the shorter row list of a step's two columns, summed over steps: reference include/plinopt_optimize.inl:92-94 walks
 * every row). */
int plo_cse_plan_hbm_counters_ex(const plo_plan_t *plan, uint32_t *out, uint32_t n);
int plo_cse_plan_destroy(plo_plan_t *plan);

/* Replaces the body of `#pragma omp parallel for` in CSEOptimiser,
 * include/plinopt_optimize.inl:1204-1238 (and the same pattern at :1056-1100,
 * :1137-1177): evaluates candidates seed0 .. seed0+nseeds-1 with the
 * per-candidate kernel Optimizer() (:616-631: OneSub :209-314, RemOneCSE
 * :60-194, ProgramGen :513-611, counts only) and returns the minimum under the
 * total order (cmpOpCount key, seed). Text of the winner is produced by the
 * host replaying `seed` (plo_cse_replay). */
int plo_cse_search_plan(plo_plan_t *plan, uint64_t seed0, uint64_t nseeds,
                        int cost_mode, plo_best_t *out, plo_stats_t *stats);
int plo_cse_search(const plo_csr_t *A, uint32_t p, uint64_t seed0, uint64_t nseeds,
                   int cost_mode, plo_best_t *out, plo_stats_t *stats);
/* The same search over `ndev` devices of one node from ONE process: the restart range in ndev contiguous shards (the blocks
 * bin/optimizer --gpu N uses), one host thread and one device per shard (devices[k], or 0..ndev-1 when devices is NULL;
 * the same ordinal may be listed twice), each thread with its own stream and plan; the result is the minimum under the
 * total order (cmpOpCount key of include/plinopt_optimize.h:53-64, seed) -- what the `#pragma omp critical` of
 * include/plinopt_optimize.inl:1214-1237 keeps.  stats->kernel_ms is the slowest shard's kernel time.  With two or more
 * DISTINCT devices the minimum is taken by RCCL over a communicator of the devices (librccl loaded at run time, the
 * communicator kept for the life of the process): one 8-byte MIN all-reduce of the cost key, one of the seed offset among the
 * shards that hold the minimal key -- the lexicographic minimum whatever the width of the costs.  The host minimum is its check:
 * a difference is a '#' diagnostic on stderr and the host value is returned.  stats->reduce says whether the all-reduce gave the
 * result, stats->reduce_seconds its wall time.  PLO_MULTI_REDUCE=host|rccl overrides (rccl: also with one device; then a missing
 * librccl is an error).  A shard the device refuses (PLO_E_CAPACITY / PLO_E_UNSUPPORTED) makes the call return that code.
 * Across processes bench.py and plinopt_amd/dist.py reduce the same words with torch.distributed (RCCL). */
int plo_cse_search_multi(const plo_csr_t *A, uint32_t p, uint64_t seed0, uint64_t nseeds, int cost_mode,
                         int ndev, const int *devices, plo_best_t *out, plo_stats_t *stats);
/* communicators built by this process so far (one per distinct device set: a second call on the same devices builds none) */
uint64_t plo_multi_comm_inits(void);

/* Same candidates, every (adds, muls) written back: the parity-test entry.
 * seeds == NULL means seed0, seed0+1, ... */
int plo_cse_cost_many_plan(plo_plan_t *plan, const uint64_t *seeds, uint64_t seed0,
                           uint64_t n, uint32_t *adds, uint32_t *muls, plo_stats_t *stats);
int plo_cse_cost_many(const plo_csr_t *A, uint32_t p, const uint64_t *seeds, uint64_t seed0,
                      uint64_t n, uint32_t *adds, uint32_t *muls);

/* Two matrices per candidate, one random stream.  Replaces the body of the restart loop of
 * LUOptimiser, include/plinopt_optimize.inl:1056-1100: Optimizer() on a copy of U (:1068) and then
 * on a copy of L (:1072), op-counts added (:1078-1079), best kept under cmpOpCount (:1081-1099).
 * Both matrices must fit the LDS-resident kernel. */
typedef struct plo_chain plo_chain_t;
int plo_cse_chain_create(const plo_csr_t *first, const plo_csr_t *second, uint32_t p, plo_chain_t **chain);
int plo_cse_chain_destroy(plo_chain_t *chain);
int plo_cse_chain_search(plo_chain_t *chain, uint64_t seed0, uint64_t nseeds, int cost_mode,
                         plo_best_t *out, plo_stats_t *stats);
int plo_cse_chain_cost_many(plo_chain_t *chain, const uint64_t *seeds, uint64_t seed0, uint64_t n,
                            uint32_t *adds, uint32_t *muls, plo_stats_t *stats);

/* The same chained evaluation for MANY pairs in one launch: candidate c (0 <= c < npairs*per_pair) runs on pair c / per_pair with
 * seed seed0 + c.  The kernel method (nullspacedecomp :689-884) has a different pair (Free, Dep) for every restart, or small
 * block of restarts; one plan + one launch per pair would leave the GPU idle.  adds/muls (npairs*per_pair entries) and best
 * may each be NULL.  With PLO_COST_SUM the split of the winner is not returned (best->adds = the sum). */
int plo_cse_chain_batch(uint32_t npairs, const plo_csr_t *firsts, const plo_csr_t *seconds, uint32_t p, uint64_t seed0, uint32_t per_pair,
                        int cost_mode, uint32_t *adds, uint32_t *muls, plo_best_t *best, plo_stats_t *stats);


/* The kernel method of bin/optimizer -K with EVERYTHING on the device (plo::kmethod_kernel): restart c (seed seed0 + c) is one
 * pass of the loop of KernelOptimiser, include/plinopt_optimize.inl:1299-1340 -- a nullspace decomposition of M
 * (nullspacedecomp :689-884: random row order, greedy row basis, NotIndep; this build's rule, restated in
 * oracle/plo_oracle.c plo_oracle_kernel_restart), then Optimizer() on Free and on Dep from one random stream (:1322-1333).
 * per_block > 1: the restarts of a block of per_block consecutive seeds share the decomposition of the block's first seed.
 * adds/muls (nrestarts entries), info (3 per restart: rank, NotIndep, number of dependent rows computed through Dep) and best may each be NULL.
 * M needs at most 128 rows, 64 columns, 64 dependent rows and a kernel of positive dimension (PLO_E_UNSUPPORTED otherwise: the caller
 * falls back to host decompositions + plo_cse_chain_batch). */
int plo_kernel_search(const plo_csr_t *M, uint32_t p, uint64_t seed0, uint64_t nrestarts, uint32_t per_block, int cost_mode,
                      uint32_t *adds, uint32_t *muls, uint32_t *info, plo_best_t *best, plo_stats_t *stats);
/* The restart loop of KernelOptimiser (include/plinopt_optimize.inl:1299-1340) over `ndev` devices from one process: contiguous
 * shards of the restart range, one host thread and one device each, minimum under (cmpOpCount key, seed) by the RCCL MIN
 * all-reduces of plo_cse_search_multi (same conventions for devices, stats->reduce and PLO_MULTI_REDUCE).  per_block must be 1
 * with more than one device.  BASELINE `bin/optimizer -K --gpu N`. */
int plo_kernel_search_multi(const plo_csr_t *M, uint32_t p, uint64_t seed0, uint64_t nrestarts, uint32_t per_block, int cost_mode,
                            int ndev, const int *devices, plo_best_t *best, plo_stats_t *stats);


/* Change-of-basis (CoB) search of bin/sparsifier: one (block,row) enumeration of `localSparsifier`,
 * include/plinopt_sparsify.inl:282-314, i.e. |coeffs|^4 calls of `testLinComb` (:167-197).  TM is the n x m
 * matrix being sparsified (dense, row major, residues mod p), Cand the n x n matrix whose rows 0..row-1 are the
 * rows already chosen; candidate (i,j,k,l) is the row w with w[offsetblock+t] = coeffs[(i,j,k,l)[t]] (positions
 * >= n dropped, :305-310).  Result: the first candidate in lexicographic (i,j,k,l) order that is independent of
 * the chosen rows and maximises (zeros(TM^T w), zeros(w)), if it beats (w0,w1) strictly; index = ((i*C+j)*C+k)*C+l. */
typedef struct {
    int32_t  zeros_v, zeros_w;   /* score of the winner (or the incoming w0,w1 when nothing beats them) */
    uint64_t index;              /* flattened (i,j,k,l) */
    uint32_t found;              /* 1 if some candidate beat (w0,w1) */
} plo_cob_best_t;
int plo_cob_search(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock,
                   const uint32_t *coeffs, uint32_t ncoeffs, uint32_t p, int32_t w0, int32_t w1,
                   plo_cob_best_t *out, plo_stats_t *stats);
/* The same enumeration restricted to the prefixes (i,j,k) first_group .. first_group+ngroups-1 (in the order of the
 * loops at include/plinopt_sparsify.inl:299-303), every l: one shard of the |coeffs|^4 candidates.  The winner of the
 * whole enumeration is the shard winner of largest (zeros_v, zeros_w), smallest index among equals (one 8-byte MAX
 * all-reduce of the packed word, plinopt_amd/dist.py). */
int plo_cob_search_range(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock,
                         const uint32_t *coeffs, uint32_t ncoeffs, uint32_t p, int32_t w0, int32_t w1,
                         uint64_t first_group, uint64_t ngroups, plo_cob_best_t *out, plo_stats_t *stats);

/* Up to 4 enumerations of the same shape (n, m, row, offsetblock) in ONE launch -- one upload, one kernel, one download: the two
 * primes of an enumeration over the rationals (bin/sparsifier without -q calls localSparsifier :282-314 once over Q; this build
 * enumerates modulo two primes and checks the winners over Q), or the shards of one enumeration.  out[k] as plo_cob_search. */
typedef struct {
    const uint32_t *TM, *Cand, *coeffs;   /* residues modulo p, as for plo_cob_search */
    uint32_t ncoeffs, p;
    int32_t  w0, w1;
} plo_cob_problem_t;
int plo_cob_search_batch(uint32_t nprob, uint32_t n, uint32_t m, uint32_t row, uint32_t offsetblock,
                         const plo_cob_problem_t *prob, plo_cob_best_t *out, plo_stats_t *stats);

/* ---- -E, the exhaustive CSE tree: RecSub / RecOptimizer (include/plinopt_optimize.inl:889-1013, called by AllCSEOpt
 * :1252-1281) explore every schedule of pairs of frequency > 1 (not only the maximal ones).  Here a schedule is a
 * candidate addressed by an index in the mixed radix of its own path: at every step the children (distinct triples of
 * frequency > 1, in std::map order) are numbered 0..T-1; digit = index mod T, index /= T.  The product of the radices met by
 * a candidate is returned; the indices first..first+count-1 = 0..N-1 cover the whole tree as soon as N >= *maxprod (the
 * largest product seen).  Cost = op-count of the program Optimizer would print for that schedule; RecSub's order is
 * PLO_COST_ADD_THEN_MUL (:958-959).  LDS-resident plans only (PLO_E_UNSUPPORTED otherwise). */
int plo_cse_enum_cost_many_plan(plo_plan_t *plan, uint64_t first, uint64_t n, uint32_t *adds, uint32_t *muls, uint64_t *prods, plo_stats_t *stats);
int plo_cse_enum_search_plan(plo_plan_t *plan, uint64_t first, uint64_t count, int cost_mode, plo_best_t *best /* .seed = index */,
                             uint64_t *maxprod, plo_stats_t *stats);

/* ---- In-place trilinear search: replaces the body of the restart loop of SearchTriLinearAlgorithm
 * (include/plinopt_inplace.inl:837-924; driver src/trilplacer.cpp:150-152).  A (m x nA), B (m x nB) and T = transpose of
 * the product matrix (m x nT), integer CSR with columns sorted per row.  One candidate (seed) = row permutation +
 * coherent row negations drawn from the seed's stream, then the oriented (variant 0) and the unoriented (variant 1)
 * in-place program of TriLinearProgram :732-806; cost = (ADD, SCA) lexicographic as :893-897, ties to the smaller
 * (seed, variant).  seed == PLO_TRIL_BASE_SEED is the unpermuted oriented program of :829.  The device path takes
 * matrices without empty row and with rows of at most 64 entries (PLO_E_UNSUPPORTED otherwise: such inputs stay on the
 * host); integer entries here, rationals through plo_tril_plan_create_q.  The host replays the winning (seed, variant) to print the program. */
#define PLO_TRIL_BASE_SEED 0xFFFFFFFFFFFFFFFFull
typedef struct { uint32_t m, n; const uint32_t *rowptr; const uint32_t *col; const int32_t *val; } plo_icsr_t;
typedef struct { uint32_t add, sca, mul; uint32_t variant; uint64_t seed; } plo_tril_best_t;
typedef struct plo_tril_plan plo_tril_plan_t;
int  plo_tril_plan_create(const plo_icsr_t *A, const plo_icsr_t *B, const plo_icsr_t *T, plo_tril_plan_t **plan);
/* expanded != 0: `trilplacer -e` (src/trilplacer.cpp:80,114-137): the program of T is TransposedDoubleAlgorithm
 * (include/plinopt_inplace.inl:507-598) on DoubleExpand(T) (:676-716, built on the device from the m-row T); a candidate
 * returns (ADD, SCA) of that variant and MUL = m double-size AXPYs (:799). */
int  plo_tril_plan_create_x(const plo_icsr_t *A, const plo_icsr_t *B, const plo_icsr_t *T, int expanded, plo_tril_plan_t **plan);
/* Rational coefficients num/den (den == NULL: all 1), as the reference's matrices are (Givaro::Rational, include/plinopt_inplace.inl:19):
 * the atoms of the device programs carry the coefficients' images modulo the 31-bit prime 2147483629 (an additive atom its signed
 * coefficient, a multiplicative atom its factor: what cumulate/isnoop/complexity, :96-144, depend on); the tool replays and checks the
 * winner over Q.  All entries +-1: the kernel of rounds 1-2.  expanded != 0 takes rational coefficients too (round 4: the scaling
 * atoms *y, *a and the atoms of z = -y c y and c of include/plinopt_inplace.inl:532-535, 561-565).  PLO_E_UNSUPPORTED: an empty row,
 * a row of more than 64 entries, or an entry that vanishes modulo the prime. */
typedef struct { uint32_t m, n; const uint32_t *rowptr; const uint32_t *col; const int64_t *num; const int64_t *den; } plo_qcsr_t;
int  plo_tril_plan_create_q(const plo_qcsr_t *A, const plo_qcsr_t *B, const plo_qcsr_t *T, int expanded, plo_tril_plan_t **plan);
void plo_tril_plan_destroy(plo_tril_plan_t *plan);
/* ops6[6k..6k+5] = ADD,SCA,MUL of variant 0 then of variant 1 for candidate k (seeds[k], or seed0+k when seeds==NULL) */
int  plo_tril_cost_many(plo_tril_plan_t *plan, const uint64_t *seeds, uint64_t seed0, uint64_t n, uint32_t *ops6, plo_stats_t *stats);
int  plo_tril_search(plo_tril_plan_t *plan, uint64_t seed0, uint64_t nseeds, plo_tril_best_t *best, plo_stats_t *stats);
/* The restart loop of SearchTriLinearAlgorithm (include/plinopt_inplace.inl:837-924) over `ndev` devices from one process --
 * BASELINE configs[3], `bin/trilplacer L R P` seed-sharded over the GPUs of a node with an RCCL MIN: contiguous shards of the seed
 * range, one host thread, one device and one plan (plo_tril_plan_create_q of the three matrices) each; the winner is the minimum
 * under (ADD, SCA) (:893-897), then (seed, variant), by the two MIN all-reduces of plo_cse_search_multi: (ADD << 32 | SCA), then
 * ((seed - seed0) << 1 | variant) among the shards holding the minimal cost.  Same conventions for devices, stats and
 * PLO_MULTI_REDUCE. */
int  plo_tril_search_multi(const plo_qcsr_t *A, const plo_qcsr_t *B, const plo_qcsr_t *T, int expanded, uint64_t seed0, uint64_t nseeds,
                           int ndev, const int *devices, plo_tril_best_t *best, plo_stats_t *stats);

/* Pack / unpack the (cost, seed) word used by the grid reduction and by the
 * single 8-byte MIN all-reduce across ranks (the `#pragma omp critical`
 * best-so-far of include/plinopt_optimize.inl:1214-1237).  seed_off is the
 * candidate's offset from seed0 (< 2^32). */
uint64_t plo_pack_cost(uint32_t adds, uint32_t muls, int cost_mode, uint32_t seed_off);

#ifdef __cplusplus
}
#endif
#endif
