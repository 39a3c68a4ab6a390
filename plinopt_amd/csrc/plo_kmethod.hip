// ===========================================================================
// plo_kmethod.hip -- the kernel method (`bin/optimizer -K`) entirely on gfx950:
// one wavefront = one restart of KernelOptimiser (reference
// include/plinopt_optimize.inl:1288-1353): a nullspace decomposition of M
// (nullspacedecomp :689-884, with this build's rule, see host/plo_host.hpp
// `kernel_decomp_order` and oracle/plo_oracle.c `plo_oracle_kernel_restart`),
// then Optimizer() on `Free` and on `Dep` from one random stream (:1322-1333).
//
// Round 1 made the decomposition and both LDS images on the host (58 ms of
// kernels in 2.6 s of wall); here the wave does it all:
//   1. copies M's image (the template of plo_cse_wave.hip) into its LDS region,
//   2. draws the row order (Fisher-Yates) and eliminates row after row: lane j
//      holds entry j of the row being reduced and, in a second register,
//      coefficient j of its combination over the basis rows found so far; the
//      echelon rows and their combinations are LDS arrays read by all lanes,
//   3. NotIndep = next() mod #dependent rows; `Free` = M's image with the kept
//      dependent rows retired from the pair table and the column masks,
//   4. Optimizer on Free (run_candidate), then the image of `Dep` is built in
//      the same region (rows = kept dependent rows, columns = rows of M, the
//      combination over the basis rows; inverses by Fermat, pair table by
//      tab_inc), Optimizer on Dep with the stream where the first call left it.
// Needs n <= 64 (a row and a combination fit one register per lane), m <= 128
// (a lane eliminates rows lane and lane + 64: `-F` on 4x4x4, [M;I] = 65 rows,
// and the 84-row P matrices of the 3x4x7 family stay here) and at most 64
// dependent rows (Dep keeps one row per lane in ProgramGen); the
// sizes of Dep's image are bounded from rank(M) by the host, its pair table by a
// sizing launch (kmethod_size_kernel); a full table is reported (ERR_TABLE) and
// the launch repeated with a larger one, as for the other wave-kernel plans.
// ===========================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

struct KPlan {
    WavePlan PM, PD;               // PM: plan of M itself (Free has its layout); PD: layout of Dep's image (no template)
    uint32_t m, n, rank, ndeps, per_block;
    uint32_t mers;                 // k when p = 2^k - 1 (shift-and-add reduction in the elimination), else 0
    uint32_t region;               // bytes of the image region of a wave (max of the two plans)
    uint32_t off_vc;               // elimination work arrays V, C inside the image region, behind M's template
    int32_t off_depc;              // the dependent rows' combinations, relative to the END of the image region: negative = inside it (round 4: behind the
                                   // elimination arrays, where Dep's tie list and multiplier list lie -- both idle until Dep's Optimizer call, by which
                                   // time the combinations have become Dep's rows: 2.1 KB less per wave on 4x4x4_49_156_L, 11 waves per CU instead of 9)
    uint32_t off_vrow, off_ord, off_piv, off_basis, off_deps, off_rsd, scratch_bytes;   // scratch behind the region
    const uint64_t *rsD;           // unused since round 3 (Dep's row starts are per restart: KScratch::rsd)
};
#ifdef PLO_KM_PROFILE
__device__ unsigned long long g_kprof[8];   // lane 0 of every wave: cycles in image copy + decomposition, Free build, Optimizer on Free, Dep build, Optimizer on Dep; restarts
#define KP_T(k_) do { const unsigned long long t__ = clock64(); kp_acc[k_] += t__ - kp_t; kp_t = t__; } while (0)
#else
#define KP_T(k_) do { } while (0)
#endif
struct KInfo { uint32_t *info; };  // optional, 3 words per candidate: rank, NotIndep, dependent rows
enum { ERR_KDEC = 5 };             // device error word: decomposition inconsistent (rank, table of M)

// product mod p for the elimination: shift-and-add when p is a Mersenne prime (131071 = 2^17 - 1, the tools' usual modulus), Barrett otherwise
__device__ __forceinline__ uint32_t kmul(uint32_t a, uint32_t b, uint32_t p, uint64_t mu, uint32_t mers) {
    if (mers) { uint64_t x = (uint64_t)a * b; x = (x & p) + (x >> mers); x = (x & p) + (x >> mers); return (uint32_t)(x >= p ? x - p : x); }
    return fmul<false>(a, b, p, mu);
}
// x^(p-2); x = +-1 are their own inverses
__device__ __forceinline__ uint32_t kinv(uint32_t x, uint32_t p, uint64_t mu) {
    if (x == 1u || x == p - 1u) return x;
    uint32_t res = 1u, bs = x;
    for (uint32_t e = p - 2u; e; e >>= 1) { if (e & 1u) res = fmul<false>(res, bs, p, mu); bs = fmul<false>(bs, bs, p, mu); }
    return res;
}

struct KScratch { uint32_t *depc, *vrow; uint16_t *ord, *piv, *basis, *deps, *rsd; };   // rsd: row starts of THIS restart's Dep (rows packed: the image is sized from sampled entry counts, not from rows x rank)
__device__ __forceinline__ KScratch kscratch(const KPlan &K, uint8_t *scr) {
    return KScratch{(uint32_t *)(scr + (ptrdiff_t)K.off_depc), (uint32_t *)(scr + K.off_vrow),
                    (uint16_t *)(scr + K.off_ord), (uint16_t *)(scr + K.off_piv), (uint16_t *)(scr + K.off_basis), (uint16_t *)(scr + K.off_deps), (uint16_t *)(scr + K.off_rsd)};
}

// Steps 1-2 and the draw of NotIndep.  M's image must be in `reg`.  Returns the number of dependent rows KEPT in Dep
// (0: zero dimensional kernel); S.deps / S.depc hold the dependent rows in elimination order and their combinations
// (indexed by basis position), S.basis the basis rows.
__device__ uint32_t kmethod_decompose(const KPlan &K, uint8_t *reg, const KScratch &S, const uint16_t *rsM, uint64_t dseed, uint32_t lane,
                                      uint32_t &notindep_out)
{
    const WavePlan &PM = K.PM;
    const uint32_t p = PM.p, m = K.m, n = K.n, R = K.rank, mers = K.mers; const uint64_t mu = PM.mu;
    const uint32_t *valM = (const uint32_t *)(reg + PM.off_val); const uint16_t *colM = (const uint16_t *)(reg + PM.off_col), *lenM = (const uint16_t *)(reg + PM.off_len);
    // RIGHT-LOOKING elimination, lanes = rows: V[i] is row i minus the combination C[i] of the basis rows found so far
    // (row_i = sum_j C[i][j] basis_j + V[i]).  When the row whose turn it is (order of the shuffle) is non-zero it becomes the next
    // basis row: it is normalised and every row still to come is updated AT ONCE, each lane its own row -- products that do not
    // depend on each other; the left-looking form reduced one row at a time through a chain of ~16 dependent steps (measured:
    // 182 k of the 240 k cycles of a decomposition).  Same arithmetic, same results.  V and C live behind M's image in the
    // wave's region, which nothing else uses before the first Optimizer call (strides n+1 and R+1: no bank conflicts).
    const uint32_t sv = n + 1u, sc = R + 1u;
    uint32_t *V = (uint32_t *)(reg + K.off_vc), *C = V + m * sv;
    uint32_t rng = 1u + (uint32_t)(splitmix64(dseed) % 2147483646ull);
    for (uint32_t row = lane; row < m; row += 64u) S.ord[row] = (uint16_t)row;
    PLO_WAVE_SYNC();
    if (lane == 0) for (uint32_t i = m; i > 1u; --i) { const uint32_t j = rng_next(rng) % i; const uint16_t t = S.ord[i - 1u]; S.ord[i - 1u] = S.ord[j]; S.ord[j] = t; }
    rng = uni32(rng);
    PLO_WAVE_SYNC();
    uint32_t mypos[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};                          // place of this lane's rows (lane, lane + 64) in the order
    for (uint32_t q = lane; q < m; q += 64u) S.piv[S.ord[q]] = (uint16_t)q;  // (piv doubles as the inverse permutation)
    PLO_WAVE_SYNC();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t row = lane + 64u * (uint32_t)h;
        if (row < m) {
            mypos[h] = S.piv[row];
            for (uint32_t j = 0; j < n; ++j) V[row * sv + j] = 0u;
            for (uint32_t j = 0; j < R; ++j) C[row * sc + j] = 0u;
            const uint32_t base = rsM[row], ln = lenM[row];
            for (uint32_t z = 0; z < ln; ++z) V[row * sv + colM[base + z]] = valM[base + z];
        }
    }
    PLO_WAVE_SYNC();
    uint32_t nb = 0, nd = 0;
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t L = uni32(S.ord[t]);
        const uint32_t vj = lane < n ? V[L * sv + lane] : 0u;
        const uint64_t nz = __ballot(vj != 0u);
        if (!nz) {                                                           // row L = sum_j C[L][j] basis_j: dependent
            if (lane == 0 && nd < K.ndeps) S.deps[nd] = (uint16_t)L;
            ++nd;
            continue;
        }
        if (nb >= R) { ++nb; continue; }                                     // (cannot happen: the rank does not depend on the order)
        const uint32_t pc = (uint32_t)__builtin_ctzll(nz);
        const uint32_t iv = kinv((uint32_t)__builtin_amdgcn_readlane((int)vj, (int)pc), p, mu);
        // the new basis row number nb: echelon row = V[L] / pivot, its combination = (-C[L] / pivot, 1 / pivot at position nb)
        if (lane < n) V[L * sv + lane] = kmul(vj, iv, p, mu, mers);
        if (lane < nb) { const uint32_t q = kmul(C[L * sc + lane], iv, p, mu, mers); C[L * sc + lane] = q ? p - q : 0u; }
        else if (lane == nb) C[L * sc + lane] = iv;
        if (lane == 0) S.basis[nb] = (uint16_t)L;
        PLO_WAVE_SYNC();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t row = lane + 64u * (uint32_t)h;
            if (row < m && mypos[h] > t) {                                   // rows still to come
                const uint32_t x = V[row * sv + pc];
                if (x) {
                    for (uint32_t j = 0; j < n; ++j) {
                        const uint32_t e = V[L * sv + j];
                        if (e) { const uint32_t xe = kmul(x, e, p, mu, mers), w = V[row * sv + j]; V[row * sv + j] = w >= xe ? w - xe : w + p - xe; }
                    }
                    for (uint32_t j = 0; j <= nb; ++j) {
                        const uint32_t cb = C[L * sc + j];
                        if (cb) { uint32_t w = C[row * sc + j] + kmul(x, cb, p, mu, mers); w -= w >= p ? p : 0u; C[row * sc + j] = w; }
                    }
                }
            }
        }
        ++nb;
        PLO_WAVE_SYNC();
    }
    notindep_out = 0;
    if (nd == 0u || nb != R || nd != K.ndeps) return 0u;
    // combinations of the dependent rows, in elimination order, for the Dep image
    for (uint32_t d = 0; d < nd; ++d) { const uint32_t L = uni32(S.deps[d]); if (lane < R) S.depc[d * R + lane] = C[L * sc + lane]; }
    PLO_WAVE_SYNC();
    const uint32_t ni = uni32(rng_next(rng) % nd);                           // :792-795
    notindep_out = ni;
    return nd - ni;
}

// length of row j of Dep (its non-zero coefficients over the basis rows)
__device__ __forceinline__ uint32_t kmethod_dep_row(const KPlan &K, const KScratch &S, uint32_t j, uint32_t lane) {
    const uint32_t R = K.rank;
    return (uint32_t)__popcll(__ballot(lane < R && S.depc[j * R + lane] != 0u));
}

// One restart.  Returns adds<<32 | muls of the two programs together.
template <bool UNITM>
__device__ uint64_t kmethod_candidate(const KPlan &K, uint8_t *reg, uint8_t *scr, const uint16_t *rsM, const uint16_t *rsD, uint64_t dseed, uint64_t seed,
                                      uint32_t lane, uint32_t *errw, uint32_t *info3)
{
    const WavePlan &PM = K.PM, &PD = K.PD;
    const uint32_t p = PM.p, m = K.m, R = K.rank; const uint64_t mu = PM.mu;
    const KScratch S = kscratch(K, scr);
#ifdef PLO_KM_PROFILE
    unsigned long long kp_t = clock64(), kp_acc[5] = {0, 0, 0, 0, 0};
#endif
    for (uint32_t i = lane; i < (PM.tmpl_bytes >> 3); i += 64u) ((uint64_t *)reg)[i] = PM.tmpl[i];
    PLO_WAVE_SYNC();
    uint32_t notindep = 0;
    const uint32_t kept = kmethod_decompose(K, reg, S, rsM, dseed, lane, notindep);
    if (info3 && lane == 0) { info3[0] = R; info3[1] = notindep; info3[2] = kept; }
    if (kept == 0u) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_KDEC); return 0; }
    KP_T(0);
    bool bad = false;
    {   // ---- Free: M's image minus the kept dependent rows (their pairs retired, their mask bits cleared)
        uint64_t *tab = (uint64_t *)(reg + PM.off_tab), *cmask = (uint64_t *)(reg + PM.off_cmask), *umask = (uint64_t *)(reg + PM.off_umask);
        uint32_t *val = (uint32_t *)(reg + PM.off_val), *inv = (uint32_t *)(reg + PM.off_inv);
        uint16_t *col = (uint16_t *)(reg + PM.off_col), *len = (uint16_t *)(reg + PM.off_len);
        const uint32_t abs_ = PM.rb + PM.bb, rb = PM.rb, msM = 2u * PM.mw;      // masks: {cmask[mw], umask[mw]} per column
        // G rows per trip (lane groups of LPR lanes, as in the sweeps of the wave kernel): lane t of a group retires the pairs its
        // entry forms with every earlier entry of the row, and clears its column's mask bits
        const uint32_t LPR = 1u << PM.lpr_log2, G = 64u >> PM.lpr_log2, g = lane >> PM.lpr_log2, t = lane & (LPR - 1u);
        for (uint32_t r0 = 0; r0 < kept; r0 += G) {
            const bool act = r0 + g < kept;
            const uint32_t i = act ? (uint32_t)S.deps[r0 + g] : 0u, base = act ? rsM[i] : 0u, ln = act ? (uint32_t)len[i] : 0u;
            const bool have = t < ln;
            const uint32_t cy = have ? col[base + t] : 0u, vy = have ? val[base + t] : 0u;
            for (uint32_t x = 0; x + 1u < PM.maxlen; ++x) {
                if (have && x < t) {
                    const uint32_t r = fmul<UNITM>(vy, UNITM ? val[base + x] : inv[base + x], p, mu);
                    bad |= !tab_dec(tab, ((uint64_t)col[base + x] << abs_) | ((uint64_t)cy << rb) | r, PM.cap, PM.hbits);
                }
            }
            if (have) { atomicAnd((unsigned long long *)&cmask[cy * msM + (i >> 6)], ~(1ull << (i & 63u))); atomicAnd((unsigned long long *)&umask[cy * msM + (i >> 6)], ~(1ull << (i & 63u))); }
            PLO_WAVE_SYNC();
            if (act && t == 0u) len[i] = 0;
        }
        PLO_WAVE_SYNC();
    }
    if (__ballot(bad)) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_KDEC); return 0; }
    KP_T(1);
    PickState ps{1u + (uint32_t)(splitmix64(seed) % 2147483646ull), 0u, 0ull, 1ull, 0u};
    const uint64_t r1 = run_candidate<UNITM>(PM, reg, rsM, ps, lane, errw);
    PLO_WAVE_SYNC();
    KP_T(2);
    {   // ---- Dep: kept rows, columns = rows of M
        uint64_t *tab = (uint64_t *)(reg + PD.off_tab), *cmask = (uint64_t *)(reg + PD.off_cmask), *umask = (uint64_t *)(reg + PD.off_umask);
        uint32_t *val = (uint32_t *)(reg + PD.off_val), *inv = (uint32_t *)(reg + PD.off_inv);
        uint16_t *col = (uint16_t *)(reg + PD.off_col), *len = (uint16_t *)(reg + PD.off_len);
        const uint32_t abs_ = PD.rb + PD.bb, rb = PD.rb;
        for (uint32_t s = lane; s < PD.cap; s += 64u) tab[s] = PLO_EMPTY;
        for (uint32_t c = lane; c < m; c += 64u) { cmask[c * 2u] = 0ull; umask[c * 2u] = 0ull; }      // (Dep has at most 64 rows: one mask word)
        if (lane < PD.m) len[lane] = 0;
        PLO_WAVE_SYNC();
        // Rows of Dep, G at a time (groups of LPR >= rank lanes).  The columns of Dep are the basis rows of M: lane s of a group
        // stands for the basis row of s-th smallest index, so a row's entries come out in increasing column order and are
        // compacted by a ballot over the group.
        {
            const uint32_t lprD = PD.lpr_log2, LPRD = 1u << lprD, GD = 64u >> lprD, gD = lane >> lprD, sD = lane & (LPRD - 1u);
            const uint64_t gmD = (LPRD == 64u) ? ~0ull : (((1ull << LPRD) - 1ull) << (gD << lprD));
            // basis position of the s-th smallest basis row (the same for every row of Dep)
            uint32_t bpos = 0, bcol = 0;
            if (lane < R) {                                                  // lane b: rank of basis row b among the basis rows -> vrow[rank] = b
                const uint32_t rowb = S.basis[lane]; uint32_t rk = 0;
                for (uint32_t b2 = 0; b2 < R; ++b2) rk += (uint32_t)S.basis[b2] < rowb ? 1u : 0u;
                S.vrow[rk] = lane;
            }
            PLO_WAVE_SYNC();
            if (sD < R) { bpos = S.vrow[sD]; bcol = S.basis[bpos]; }
            // pass 1: the lengths; then the row starts of this restart (rows packed one behind the other)
            for (uint32_t r0 = 0; r0 < kept; r0 += GD) {
                const uint32_t j = r0 + gD; const bool act = j < kept && sD < R;
                const bool has = act && S.depc[j * R + bpos] != 0u;
                const uint64_t mk = __ballot(has) & gmD;
                if (j < kept && sD == 0u) len[j] = (uint16_t)__popcll(mk);
            }
            PLO_WAVE_SYNC();
            {
                uint32_t l = lane < kept ? (uint32_t)len[lane] : 0u, inc = l;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += u; }
                if (lane < kept) S.rsd[lane + 1u] = (uint16_t)inc;
                if (lane == 0u) S.rsd[0] = 0;
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if (total > PD.nnz) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_TABLE); return 0; }      // more entries than the sampled bound: the host repeats with the hard one
            }
            PLO_WAVE_SYNC();
            for (uint32_t r0 = 0; r0 < kept; r0 += GD) {
                const uint32_t j = r0 + gD; const bool act = j < kept && sD < R;
                const uint32_t x = act ? S.depc[j * R + bpos] : 0u;
                const bool has = x != 0u;
                const uint64_t mk = __ballot(has) & gmD;
                if (has) {
                    const uint32_t pos = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull)), base = S.rsd[j];
                    col[base + pos] = (uint16_t)bcol; val[base + pos] = x;
                    atomicOr((unsigned long long *)&cmask[bcol * 2u], 1ull << j);
                    if (absone(x, p)) atomicOr((unsigned long long *)&umask[bcol * 2u], 1ull << j);
                }
            }
        }
        PLO_WAVE_SYNC();
        for (uint32_t idx = lane; idx < (uint32_t)S.rsd[kept]; idx += 64u) inv[idx] = kinv(val[idx], p, mu);      // inverses of all entries, 64 at a time
        PLO_WAVE_SYNC();
        // pair table (listpairs :30-41): G rows per trip, lane t of a group pairs entry t with every earlier entry x
        const uint32_t LPR = 1u << PD.lpr_log2, G = 64u >> PD.lpr_log2, g = lane >> PD.lpr_log2, t = lane & (LPR - 1u);
        for (uint32_t r0 = 0; r0 < kept; r0 += G) {
            const uint32_t row = r0 + g; const bool act = row < kept;
            const uint32_t base = act ? (uint32_t)S.rsd[row] : 0u, ln = act ? len[row] : 0u;
            const bool have = t < ln;
            const uint32_t cy = have ? col[base + t] : 0u, vy = have ? val[base + t] : 0u;
            for (uint32_t x = 0; x + 1u < R; ++x) {
                if (have && x < t) {
                    const uint32_t r = fmul<false>(vy, inv[base + x], p, mu);
                    bad |= !tab_inc(tab, ((uint64_t)col[base + x] << abs_) | ((uint64_t)cy << rb) | r, PD.cap, PD.hbits);
                }
            }
        }
        PLO_WAVE_SYNC();
    }
    if (__ballot(bad)) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_TABLE); return 0; }
    KP_T(3);
    const uint64_t r2 = run_candidate<false>(PD, reg, S.rsd, ps, lane, errw);
    KP_T(4);
#ifdef PLO_KM_PROFILE
    if (lane == 0) { for (int q_ = 0; q_ < 5; ++q_) atomicAdd(&g_kprof[q_], kp_acc[q_]); atomicAdd(&g_kprof[5], 1ull); }
#endif
    return ((uint64_t)((uint32_t)(r1 >> 32) + (uint32_t)(r2 >> 32)) << 32) | ((uint32_t)r1 + (uint32_t)r2);
}

__device__ __forceinline__ void kmethod_stage_rs(const KPlan &K, uint64_t *lds64) {
    const uint32_t rwM = K.PM.rs_bytes >> 3, rwD = K.PD.rs_bytes >> 3, twM = K.PM.tmpl_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < rwM; i += blockDim.x) lds64[i] = K.PM.tmpl[twM + i];
    if (K.rsD) for (uint32_t i = threadIdx.x; i < rwD; i += blockDim.x) lds64[rwM + i] = K.rsD[i];
    __syncthreads();
}

template <bool UNITM>
__global__ __launch_bounds__(256) void kmethod_kernel(KPlan K, WaveJob J, KInfo I)
{
    extern __shared__ uint64_t lds64[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    kmethod_stage_rs(K, lds64);
    const uint16_t *rsM = (const uint16_t *)lds64, *rsD = (const uint16_t *)(lds64 + (K.PM.rs_bytes >> 3));
    uint8_t *reg = (uint8_t *)lds64 + K.PM.rs_bytes + K.PD.rs_bytes + (size_t)wave * (K.region + K.scratch_bytes);
    uint8_t *scr = reg + K.region;
    uint64_t best = ~0ull;
    const uint64_t stride = (uint64_t)gridDim.x * nwaves;
    for (uint64_t c = (uint64_t)blockIdx.x * nwaves + wave; c < J.ncand; c += stride) {
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        const uint64_t dseed = (!J.seeds && K.per_block > 1u) ? J.seed0 + (c / K.per_block) * K.per_block : seed;   // one decomposition per block of restarts
        const uint64_t res = kmethod_candidate<UNITM>(K, reg, scr, rsM, rsD, dseed, seed, lane, J.err, I.info ? I.info + 3ull * c : nullptr);
        const uint32_t a = (uint32_t)(res >> 32), mu_ = (uint32_t)res;
        if (lane == 0) { if (J.adds) J.adds[c] = a; if (J.muls) J.muls[c] = mu_; }
        const uint64_t packed = ((uint64_t)cost_key32(a, mu_, J.cost_mode) << 32) | (uint32_t)c;
        best = packed < best ? packed : best;
        PLO_WAVE_SYNC();
    }
    if (J.best) {
        __syncthreads();
        if (lane == 0) lds64[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t b = lds64[0];
            for (uint32_t w = 1; w < nwaves; ++w) b = lds64[w] < b ? lds64[w] : b;
            if (b != ~0ull) atomicMin(J.best, (unsigned long long)b);
        }
    }
}
template __global__ void kmethod_kernel<true>(KPlan, WaveJob, KInfo);
template __global__ void kmethod_kernel<false>(KPlan, WaveJob, KInfo);

// Sizing launch: the decompositions of a sample of the restarts; sz[0] = max over them of the number of pairs of Dep
// (an upper bound of its distinct triples), sz[1] = max entries of Dep, sz[2] = 1 when a decomposition found no dependent row.
__global__ __launch_bounds__(256) void kmethod_size_kernel(KPlan K, WaveJob J, uint32_t *sz)
{
    extern __shared__ uint64_t lds64[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t rwM = K.PM.rs_bytes >> 3, twM = K.PM.tmpl_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < rwM; i += blockDim.x) lds64[i] = K.PM.tmpl[twM + i];
    __syncthreads();
    const uint16_t *rsM = (const uint16_t *)lds64;
    uint8_t *reg = (uint8_t *)lds64 + K.PM.rs_bytes + (size_t)wave * (K.region + K.scratch_bytes);
    const KScratch S = kscratch(K, reg + K.region);
    const uint64_t stride = (uint64_t)gridDim.x * nwaves;
    for (uint64_t c = (uint64_t)blockIdx.x * nwaves + wave; c < J.ncand; c += stride) {
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        const uint64_t dseed = (!J.seeds && K.per_block > 1u) ? J.seed0 + (c / K.per_block) * K.per_block : seed;
        for (uint32_t i = lane; i < twM; i += 64u) ((uint64_t *)reg)[i] = K.PM.tmpl[i];
        PLO_WAVE_SYNC();
        uint32_t ni = 0;
        const uint32_t kept = kmethod_decompose(K, reg, S, rsM, dseed, lane, ni);
        if (kept == 0u) { if (lane == 0) atomicMax(&sz[2], 1u); continue; }
        uint32_t pairs = 0, ent = 0;
        for (uint32_t j = 0; j < kept; ++j) {
            const uint32_t ln = kmethod_dep_row(K, S, j, lane);
            pairs += ln * (ln - (ln ? 1u : 0u)) / 2u; ent += ln;
        }
        if (lane == 0) { atomicMax(&sz[0], pairs); atomicMax(&sz[1], ent); }
        PLO_WAVE_SYNC();
    }
}

} // namespace plo
