// ===========================================================================
// plo_capi.hip -- host side of libplinopt_hip.so: the C-ABI declared in
// include/plinopt_hip.h.  Prepares a matrix once (row image, modular inverses,
// initial pair table and column bit-masks: the state OneSub builds at
// reference include/plinopt_optimize.inl:214-225 is identical for every
// restart, so it is computed once on the host and kept resident in HBM), then
// launches the per-candidate wave kernel over a seed range.
// There is NO CPU fallback in this file: without a HIP device every compute
// entry point fails with PLO_E_HIP.
// ===========================================================================
#include "plo_cse_wave.hip"
#include "plo_kmethod.hip"
#include "plo_cse_big.hip"
#include "plo_cob.hip"
#include "plo_tril.hip"
#include "../../include/plinopt_hip.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#ifndef PLO_BIG_LOGTRIG_MAX
// log records between two merges of the deferred updates, at most (the partitions' capacity -- what a merge can sum in its LDS table --
// is the other bound: 5.6e6 on config 5).  Round 4: 8 M instead of 4 M: 7 merges per candidate instead of 8 (one forced by the log
// instead of two), 789 -> 811 candidates/s on one box (profiles/r04_ab_config5.txt); 11 MB more workspace per candidate.
#define PLO_BIG_LOGTRIG_MAX (8ull << 20)
#endif

namespace {

thread_local std::string g_err;
// The device context of the calling thread: the process-wide one set by plo_init (one process per GPU is the model of the
// tools and of bench.py), or a thread's own inside plo_cse_search_multi (one host thread per device).
struct DevCtx {
    int device = -1; hipStream_t stream = nullptr; int cus = 0; size_t lds_max = 0;
    // scratch of the change-of-basis enumeration, kept between calls (bin/sparsifier -c 4 makes dozens of 256-candidate enumerations:
    // four allocations, four frees and two event objects per call cost more than the kernel)
    uint32_t *cob_buf = nullptr; size_t cob_words = 0; hipEvent_t cob_e0 = nullptr, cob_e1 = nullptr;
};
DevCtx g_ctx0;
thread_local DevCtx *t_ctx = nullptr;
inline DevCtx &cur_ctx() { return t_ctx ? *t_ctx : g_ctx0; }
#define g_device (cur_ctx().device)
#define g_stream (cur_ctx().stream)
#define g_cus (cur_ctx().cus)
#define g_lds_max (cur_ctx().lds_max)

int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(PLO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

uint32_t inv_mod(uint32_t a, uint32_t p) {
    int64_t t = 0, nt = 1, r = p, nr = a % p;
    while (nr) { int64_t q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; }
    if (t < 0) t += p;
    return (uint32_t)t;
}
uint32_t ceil_log2(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) ++l; return l; }
uint32_t round_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

} // namespace

// what is wrong with a caller's CSR matrix over Z_p, or nullptr (the checks every entry point makes before the host
// images are built from it: the builders index by column and invert every value)
static const char *csr_defect(const plo_csr_t *A, uint32_t p)
{
    if (!A || !A->rowptr || (A->rowptr[A->m] && (!A->col || !A->val))) return "null matrix arrays";
    for (uint32_t i = 0; i < A->m; ++i) {
        if (A->rowptr[i + 1] < A->rowptr[i]) return "rowptr not monotone";
        for (uint32_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
            if (A->col[k] >= A->n) return "column index out of range";
            if (k > A->rowptr[i] && A->col[k] <= A->col[k - 1]) return "columns must be strictly increasing inside a row";
            if (A->val[k] == 0 || A->val[k] >= p) return "values must be canonical non-zero residues";
        }
    }
    return nullptr;
}

struct plo_plan {
    plo::WavePlan P{};
    void *d_tmpl = nullptr;
    uint32_t *d_err = nullptr;
    unsigned long long *d_best = nullptr;
    uint32_t waves_per_wg = 0, lds_bytes = 0, blocks_per_cu = 0;
    uint64_t algo_bytes = 0, pairs0 = 0, distinct0 = 0;
    uint32_t p = 0;
    // host copy of the CSR for re-planning with a larger table
    std::vector<uint32_t> rowptr, col, val;
    uint32_t m = 0, n = 0;
    uint32_t cap_scale = 2;
    // HBM-resident variant (one workgroup per candidate)
    bool big = false;
    plo::BigPlan B{};
    std::vector<void *> big_bufs;          // shared immutable device buffers
    bool big_no_defer = false;             // a launch with deferred updates ran out of room in one of their structures: this plan keeps the eager table
    uint32_t big_refits = 0;               // how often that happened (plo_cse_plan_hbm_counters)
    uint32_t big_cap_scale = 1;            // eager table: slots = scale x (2 x initial triples of frequency >= 2), x4 whenever a candidate fills it
    void *d_ws = nullptr; uint64_t ws_slices = 0;
    unsigned long long *d_next = nullptr; uint32_t *d_stats = nullptr;
    uint32_t big_lds = 0;
};

namespace {

// img_out != nullptr: host part only (the template image and the plan fields; P.tmpl is left for the caller)
int build_plan(plo_plan *pl, std::vector<uint8_t> *img_out = nullptr)
{
    const uint32_t m = pl->m, n = pl->n, p = pl->p;
    const auto &rowptr = pl->rowptr; const auto &col = pl->col; const auto &val = pl->val;
    const uint32_t nnz = rowptr[m];
    plo::WavePlan &P = pl->P;
    P = plo::WavePlan{};
    uint32_t maxlen = 0; bool unit = true;
    for (uint32_t i = 0; i < m; ++i) maxlen = std::max(maxlen, rowptr[i + 1] - rowptr[i]);
    for (uint32_t k = 0; k < nnz; ++k) unit = unit && (val[k] == 1u % p || val[k] == p - 1);
    if (maxlen > 64) return fail(PLO_E_CAPACITY, "row longer than 64 entries: not handled by the LDS-resident wave kernel");
    if (m == 0 || m > 31 * 64) return fail(PLO_E_CAPACITY, "row count outside [1,1984] for the wave kernel");
    const uint32_t mw = (m + 63) / 64;
    if (!unit && mw > 1) return fail(PLO_E_CAPACITY, "more than 64 rows with non +-1 coefficients: ProgramGen of the wave kernel keeps one row per lane");   // -> HBM family
    // every CSE step lowers sum(len-1) by its frequency >= 2, so there are at most naive_adds/2 steps
    uint32_t naive = 0;
    for (uint32_t i = 0; i < m; ++i) { uint32_t l = rowptr[i + 1] - rowptr[i]; if (l > 1) naive += l - 1; }
    const uint64_t NC = (uint64_t)n + naive / 2 + 2;
    const uint32_t rb = ceil_log2(p), bb = ceil_log2((uint32_t)NC);     // residues < 2^rb, columns < 2^bb
    if (NC >= 0xFFFFull || 2u * bb + rb > 51u)
        return fail(PLO_E_CAPACITY, "pair key (col,col,ratio) does not fit 51 bits for this matrix/modulus");
    if (2ull * nnz >= 65535ull) return fail(PLO_E_CAPACITY, "op-count may exceed 16 bits");

    std::vector<uint32_t> inv(nnz);
    for (uint32_t k = 0; k < nnz; ++k) inv[k] = inv_mod(val[k], p);
    // initial pair table (listpairs, plinopt_optimize.inl:30-41; PairMap :220-225)
    std::map<uint64_t, uint32_t> pm; uint64_t pairs0 = 0;
    for (uint32_t i = 0; i < m; ++i)
        for (uint32_t x = rowptr[i]; x < rowptr[i + 1]; ++x)
            for (uint32_t y = x + 1; y < rowptr[i + 1]; ++y) {
                uint32_t r = (uint32_t)((uint64_t)val[y] * inv[x] % p);
                uint64_t key = ((uint64_t)col[x] << (bb + rb)) | ((uint64_t)col[y] << rb) | r;
                pm[key]++; ++pairs0;
            }
    pl->pairs0 = pairs0; pl->distinct0 = pm.size();
    pl->algo_bytes = 8ull * nnz + 12ull * pairs0 + 8ull;   // B_cand, SURVEY.md 8(d)
    uint32_t cap = 64;
    while (cap < (pl->cap_scale * (uint32_t)pm.size() * 3u) / 4u + 16u) cap <<= 1;   // 1.5x the initial triples (cap_scale starts at 2); a full table is detected on the device and the launch repeated with more
    const uint32_t hbits = ceil_log2(cap);

    P.m = m; P.n = n; P.nnz = nnz; P.p = p; P.NC = (uint32_t)NC; P.cap = cap; P.hbits = hbits;
    P.lpr_log2 = std::max(2u, ceil_log2(std::max(maxlen, 1u))); P.mw = mw; P.unit = unit ? 1u : 0u;
    P.multcap = unit ? 0u : (uint32_t)(naive / 2 + 8); P.maxlen = maxlen; P.rb = rb; P.bb = bb;
    if (cap > 65536u) return fail(PLO_E_CAPACITY, "pair table larger than 65536 slots");
    P.mu = p ? (~0ull) / p : 0;
    uint32_t off = 0;
    P.off_tab = off;   off += cap * 8u;
    P.off_val = off;   off += nnz * 4u;
    P.off_inv = off;   off += unit ? 0u : nnz * 4u;
    P.off_col = off;   off += nnz * 2u;
    P.off_len = off;   off += m * 2u;
    off = round_up(off, 8);
    P.off_cmask = off;                      // interleaved {cmask[mw], umask[mw]} per column
    P.off_umask = off + mw * 8u;
    const uint32_t tmpl_bytes = off + n * 2u * mw * 8u;
    off += (uint32_t)NC * 2u * mw * 8u;
    P.tmpl_bytes = tmpl_bytes;
    P.off_aff = off;   off += (2u * mw + 1u) * 8u;
    P.off_ties = off;  off += cap * 2u;
    off = round_up(off, 8);
    P.off_mult = off;  off += P.multcap * 8u;
    P.region_bytes = round_up(off, 16);
    P.rs_bytes = round_up((m + 1) * 2u, 16);

    // template image
    std::vector<uint8_t> img(tmpl_bytes + P.rs_bytes, 0);
    uint64_t *tab = (uint64_t *)(img.data() + P.off_tab);
    for (uint32_t s = 0; s < cap; ++s) tab[s] = PLO_EMPTY;               // empty: key all ones, count 0
    for (const auto &kv : pm) {
        uint32_t x = (uint32_t)kv.first ^ ((uint32_t)(kv.first >> 32) * 0x85EBCA6Bu);
        uint32_t s = (x * 0x9E3779B1u) >> (32u - hbits);                   // == plo::tab_hash
        while (tab[s] != PLO_EMPTY) s = (s + 1) & (cap - 1);
        tab[s] = (kv.first << PLO_VB) | kv.second;
    }
    uint32_t *tv = (uint32_t *)(img.data() + P.off_val), *ti = (uint32_t *)(img.data() + P.off_inv);
    uint16_t *tc = (uint16_t *)(img.data() + P.off_col), *tl = (uint16_t *)(img.data() + P.off_len);
    uint64_t *tm = (uint64_t *)(img.data() + P.off_cmask);
    uint16_t *rs = (uint16_t *)(img.data() + tmpl_bytes);
    for (uint32_t k = 0; k < nnz; ++k) { tv[k] = val[k]; if (!unit) ti[k] = inv[k]; tc[k] = (uint16_t)col[k]; }
    for (uint32_t i = 0; i < m; ++i) {
        tl[i] = (uint16_t)(rowptr[i + 1] - rowptr[i]); rs[i] = (uint16_t)rowptr[i];
        for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            tm[(col[k] * 2u) * mw + (i >> 6)] |= 1ull << (i & 63);
            if (val[k] == 1u % p || val[k] == p - 1) tm[(col[k] * 2u + 1u) * mw + (i >> 6)] |= 1ull << (i & 63);
        }
    }
    rs[m] = (uint16_t)nnz;

    // waves per workgroup / LDS
    // waves (= candidates) per workgroup: the choice that puts most waves on a CU
    if (P.rs_bytes + P.region_bytes > g_lds_max)
        return fail(PLO_E_CAPACITY, "candidate state does not fit the 160 KiB LDS of one CU");
    uint32_t W = 1, bestw = 0;
    for (uint32_t w : {4u, 2u, 1u}) {
        const uint32_t lds = P.rs_bytes + w * P.region_bytes;
        if (lds > g_lds_max) continue;
        const uint32_t waves = std::min<uint32_t>(32u, (uint32_t)(g_lds_max / lds) * w);
        if (waves > bestw) { bestw = waves; W = w; }
    }
    pl->waves_per_wg = W; pl->lds_bytes = P.rs_bytes + W * P.region_bytes;

    if (img_out) { *img_out = std::move(img); return PLO_OK; }
    if (pl->d_tmpl) { (void)hipFree(pl->d_tmpl); pl->d_tmpl = nullptr; }
    HIPCHK(hipMalloc(&pl->d_tmpl, img.size()));
    HIPCHK(hipMemcpy(pl->d_tmpl, img.data(), img.size(), hipMemcpyHostToDevice));
    P.tmpl = (const uint64_t *)pl->d_tmpl;
    if (!pl->d_err) HIPCHK(hipMalloc((void **)&pl->d_err, sizeof(uint32_t)));
    if (!pl->d_best) HIPCHK(hipMalloc((void **)&pl->d_best, sizeof(unsigned long long)));

    const void *fn = unit ? (const void *)plo::cse_wave_kernel<true> : (const void *)plo::cse_wave_kernel<false>;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    int nb = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (int)(W * 64), pl->lds_bytes));
    pl->blocks_per_cu = (uint32_t)std::max(nb, 1);
    return PLO_OK;
}


template <class T> int upload(plo_plan *pl, const std::vector<T> &v, const T **dst)
{
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T) + 64));      // slack: the kernel copies in 16-byte units
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    pl->big_bufs.push_back(d);
    *dst = (const T *)d;
    return PLO_OK;
}

const void *big_kernel_fn(const plo::BigPlan &B)
{
    if (B.idk) return B.defer ? (const void *)plo::cse_big_kernel<2, true, true> : (const void *)plo::cse_big_kernel<2, false, true>;
    if (B.defer) return B.mode == 2u ? (const void *)plo::cse_big_kernel<2, true> : B.mode == 1u ? (const void *)plo::cse_big_kernel<1, true> : (const void *)plo::cse_big_kernel<0, true>;
    return B.mode == 2u ? (const void *)plo::cse_big_kernel<2, false> : B.mode == 1u ? (const void *)plo::cse_big_kernel<1, false> : (const void *)plo::cse_big_kernel<0, false>;
}

// Plan for the HBM-resident kernel family (plo_cse_big.hip)
int build_big_plan(plo_plan *pl)
{
    const uint32_t m = pl->m, n = pl->n, p = pl->p;
    const auto &rowptr = pl->rowptr; const auto &col = pl->col; const auto &val = pl->val;
    const uint32_t nnz = rowptr[m];
    plo::BigPlan &B = pl->B; B = plo::BigPlan{};
    if (m == 0) return fail(PLO_E_CAPACITY, "empty matrix");
    uint32_t maxlen = 1, naive = 0; bool unit = true;
    for (uint32_t i = 0; i < m; ++i) { uint32_t l = rowptr[i + 1] - rowptr[i]; maxlen = std::max(maxlen, l); if (l > 1) naive += l - 1; }
    for (uint32_t k = 0; k < nnz; ++k) unit = unit && (val[k] == 1u || val[k] == p - 1);
    // The values of a candidate are the distinct values of the input (a CSE step moves values, it creates none): rows are
    // packed as column | +-1 flag << 15 | value index << 16, with one {value, inverse} table.
    std::vector<uint32_t> dv(val.begin(), val.end());
    std::sort(dv.begin(), dv.end()); dv.erase(std::unique(dv.begin(), dv.end()), dv.end());
    if (dv.size() > 65536) return fail(PLO_E_CAPACITY, "more than 65536 distinct coefficients: the packed row entry holds a 16-bit value index");
    // at most 32 values: the <= 1024 ratios v_i/v_j get identifiers (kernel mode 2: 6-byte aggregation entries, no product in the sweep)
    const bool ratio_ids = dv.size() <= 32 && !getenv("PLO_BIG_VT_GLOBAL") && !getenv("PLO_BIG_NORID");
    std::vector<uint32_t> rat, rv;
    if (ratio_ids) {
        const uint32_t nv = (uint32_t)dv.size();
        rat.resize(nv * nv);
        for (uint32_t i = 0; i < nv; ++i) { const uint32_t ii = inv_mod(dv[i], p); for (uint32_t j = 0; j < nv; ++j) rat[j * nv + i] = (uint32_t)((uint64_t)dv[j] * ii % p); }   // rat[i * nv + j] = v_i / v_j
        rv = rat; std::sort(rv.begin(), rv.end()); rv.erase(std::unique(rv.begin(), rv.end()), rv.end());
    }
    // A pair key is (column, column, ratio) in 48 bits.  Ratio field: the residue (rb bits) -- or, when that leaves too few bits for the
    // columns (a 31-bit prime: 8 bits, 256 columns) and the matrix has ratio identifiers, the IDENTIFIER (rank in the sorted list of
    // ratios: keys keep their order), kb <= 10 bits: any modulus below 2^31 with 32768 columns (plo::cse_big_idkey_kernel).
    const uint32_t rb = ceil_log2(p);
    uint64_t NC = (uint64_t)n + naive / 2 + 2;
    uint32_t kb = rb; bool idk = false;
    {
        const uint32_t bbres = rb <= 44u ? (48u - rb) / 2u : 0u;            // column bits beside a residue
        const bool fits = rb <= 30u && bbres >= 2u && n + 2ull <= (1ull << bbres) && std::min<uint64_t>(NC, 32768) <= (1ull << bbres);
        if ((!fits || getenv("PLO_BIG_IDKEYS")) && ratio_ids && rb <= 31u) { idk = true; kb = std::max(1u, ceil_log2((uint32_t)rv.size())); }   // (PLO_BIG_IDKEYS: test knob, identifiers although residues would fit)
        else if (rb > 30 || bbres < 2u) return fail(PLO_E_CAPACITY, "modulus too large for the 48-bit pair key (and more than 32 distinct coefficients: no ratio identifiers)");
    }
    const uint32_t bbmax = std::min(15u, (48u - kb) / 2u);
    if (n + 2ull > (1ull << bbmax) || n + 2ull > 32768ull) return fail(PLO_E_CAPACITY, "too many columns for the HBM-resident kernel (48-bit pair key / 32768 columns)");
    NC = std::min<uint64_t>(std::min<uint64_t>(NC, 1ull << bbmax), 32768);   // 32768 = 512 LDS block sums x 64; exceeding it at run time is reported by the device (BERR_COLS)
    const uint32_t bb = ceil_log2((uint32_t)NC);
    if (m >= 0x7FFFu) return fail(PLO_E_CAPACITY, "more than 32766 rows: frequency does not fit the table slot");
    if (maxlen > 8192) return fail(PLO_E_CAPACITY, "row longer than 8192 entries");

    std::vector<uint2> vt(dv.size());
    for (size_t k = 0; k < dv.size(); ++k) vt[k] = make_uint2(dv[k], inv_mod(dv[k], p));
    std::vector<uint32_t> inv(nnz), ent(nnz), tptr(n + 1, 0), trows(nnz), ucount(n, 0);
    for (uint32_t k = 0; k < nnz; ++k) {
        const uint32_t vi = (uint32_t)(std::lower_bound(dv.begin(), dv.end(), val[k]) - dv.begin());
        const bool u1 = val[k] == 1u || val[k] == p - 1;
        inv[k] = vt[vi].y; ent[k] = col[k] | (u1 ? 0x8000u : 0u) | (vi << 16);
        ++tptr[col[k] + 1]; if (u1) ++ucount[col[k]];
    }
    for (uint32_t c = 0; c < n; ++c) tptr[c + 1] += tptr[c];
    { std::vector<uint32_t> pos(tptr.begin(), tptr.end() - 1);
      for (uint32_t i = 0; i < m; ++i) for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) trows[pos[col[k]]++] = i; }
    // distinct pair triples, per first column (listpairs :30-41, PairMap :220-225)
    std::vector<uint64_t> keys; std::vector<uint32_t> cnts; uint64_t pairs0 = 0; uint32_t maxf = 0;
    {
        std::vector<uint64_t> tmp;
        for (uint32_t a = 0; a < n; ++a) {
            tmp.clear();
            for (uint32_t q = tptr[a]; q < tptr[a + 1]; ++q) {
                const uint32_t i = trows[q]; uint32_t x = rowptr[i];
                while (col[x] != a) ++x;
                for (uint32_t y = x + 1; y < rowptr[i + 1]; ++y)
                {
                    uint32_t rr = (uint32_t)((uint64_t)val[y] * inv[x] % p);
                    if (idk) rr = (uint32_t)(std::lower_bound(rv.begin(), rv.end(), rr) - rv.begin());
                    tmp.push_back(((uint64_t)a << (bb + kb)) | ((uint64_t)col[y] << kb) | rr);
                }
            }
            pairs0 += tmp.size();
            std::sort(tmp.begin(), tmp.end());
            for (size_t k = 0; k < tmp.size();) {
                size_t j = k; while (j < tmp.size() && tmp[j] == tmp[k]) ++j;
                keys.push_back(tmp[k]); cnts.push_back((uint32_t)(j - k)); maxf = std::max(maxf, (uint32_t)(j - k)); k = j;
            }
        }
    }
    pl->pairs0 = pairs0; pl->distinct0 = keys.size();
    pl->algo_bytes = 8ull * nnz + 16ull * keys.size();       // distinct-triple form of B_cand for HBM-resident candidates
    B.prune = getenv("PLO_BIG_NOPRUNE") ? 0u : 1u;
    B.fwin = 2048u; if (const char *e = getenv("PLO_BIG_FWIN")) B.fwin = (uint32_t)std::min<long>(2048, std::max<long>(64, strtol(e, nullptr, 10) / 64 * 64));   // test knob: windows of the flat sweep (a multiple of 64 entries)
    if (B.prune) {   // triples of frequency 1 are never chosen and never grow: they are not kept (plo_cse_big.hip, "pruning")
        size_t w = 0;
        for (size_t k = 0; k < keys.size(); ++k) if (cnts[k] >= 2u) { keys[w] = keys[k]; cnts[w] = cnts[k]; ++w; }
        keys.resize(w); cnts.resize(w);
    }
    const uint32_t multcap = (uint32_t)std::min<uint64_t>((uint64_t)naive / 2 + 8, NC);
    uint64_t cap = 1024;
    // load <= 0.5 at the start: the retirement of a pruned triple probes to the first empty slot, and dead slots are not empty
    while (cap < (uint64_t)pl->big_cap_scale * (2ull * keys.size() + 1024ull) || cap < 2ull * nnz + 2ull * multcap + 64ull) cap <<= 1;
    if (const char *e = getenv("PLO_BIG_HBITS")) { const long hb = strtol(e, nullptr, 10); if (hb >= 10 && hb <= 30 && (1ull << hb) > keys.size() + keys.size() / 8) cap = 1ull << hb; }   // experiment knob
    const uint32_t hbits = ceil_log2((uint32_t)std::min<uint64_t>(cap, 1ull << 31));
    if (cap > (1ull << 30)) return fail(PLO_E_CAPACITY, "pair table above 2^30 slots");
    std::vector<uint32_t> hist(maxf + 2, 0);
    for (size_t k = 0; k < keys.size(); ++k) ++hist[cnts[k]];
    B.m = m; B.n = n; B.nnz = nnz; B.p = p; B.NCmax = (uint32_t)NC; B.hbits = hbits; B.rb = rb; B.bb = bb; B.kb = kb; B.idk = idk ? 1u : 0u; B.unit = unit ? 1u : 0u;
    B.maxf0 = maxf + 1; B.M0 = maxf; B.multcap = multcap; B.scr_stride = maxlen;
    B.dmcap = (uint32_t)std::min<uint64_t>(1u << 20, cap); B.hlcap = (uint32_t)std::min<uint64_t>(1u << 18, cap);   // window list: <= hlcap/2 keys per window, ping-pong halves
    B.mu = (~0ull) / p;
    B.mers = 0; for (uint32_t k = 2; k < 31; ++k) if (p == (1u << k) - 1u) B.mers = k;     // Mersenne modulus: shift-and-add reduction
    // Deferred cold updates (plo_cse_big.hip, "Deferred cold updates"): partitioned store + log + hot table instead of the one big table.
    // Taken whenever its LDS budget allows (a partition and its share of the log are summed in a 2^13-slot LDS table); PLO_BIG_EAGER=1
    // keeps the eager table (the A/B switch of the tests).
    B.defer = 0u;
    uint64_t topsum = 0;                                      // entries of the M0 longest rows: a step rewrites at most M0 rows
    { std::vector<uint32_t> ls(m); for (uint32_t i = 0; i < m; ++i) ls[i] = rowptr[i + 1] - rowptr[i];
      std::sort(ls.begin(), ls.end(), std::greater<uint32_t>());
      for (uint32_t i = 0; i < m && i < maxf; ++i) topsum += ls[i]; }
    std::vector<uint64_t> st0; std::vector<uint32_t> pc0;
    if (B.prune && !getenv("PLO_BIG_EAGER") && !pl->big_no_defer) {
        uint32_t pbits = 0; while ((keys.size() >> pbits) > 1280u && pbits < 11u) ++pbits;
        const uint32_t Pn = 1u << pbits;
        pc0.assign(Pn, 0u);
        auto part = [&](uint64_t k) { return pbits ? (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64u - pbits)) : 0u; };   // == plo::dpart
        for (uint64_t k : keys) ++pc0[part(k)];
        const uint32_t maxfill = *std::max_element(pc0.begin(), pc0.end());
        const uint32_t capp = (maxfill + maxfill / 4u + 64u + 1u) & ~1u;
        const uint64_t hotmax = std::min<uint64_t>(1ull << 17, pairs0);                 // triples alive at any time <= pair instances of the input
        const uint64_t stepmax = 3ull * topsum + 3ull * 8192ull;                        // records one step can write
        if (capp <= 5000u) {
            const uint64_t lpp = 6080u - capp;                                            // records of a partition's log that still fit the merge beside its live triples (12 records per thread, an LDS table of 2^13 slots)
            uint64_t budget = lpp * Pn * 5ull / 6ull;                                      // (hash imbalance of the partitions' shares)
            if (budget > stepmax + hotmax + 4096ull) {
                uint64_t trig = std::min<uint64_t>(budget - stepmax - hotmax - 4096ull, (uint64_t)PLO_BIG_LOGTRIG_MAX);
                if (const char *e = getenv("PLO_BIG_LOGTRIG")) trig = std::min<uint64_t>(trig, std::max<uint64_t>(1, strtoull(e, nullptr, 10)));   // test knob: merges forced by the log
                const uint64_t logcap = trig + stepmax + hotmax + 4096ull;
                B.defer = 1u; B.pbits = pbits; B.capp = capp; B.logtrig = (uint32_t)trig; B.logcap = (uint32_t)logcap;
                B.plcap = (uint32_t)std::min<uint64_t>((logcap * 6ull / 5ull + Pn - 1) / Pn + 64ull, lpp + 64ull);
                B.hwin = 8192u; if (const char *e = getenv("PLO_BIG_HWIN")) B.hwin = (uint32_t)std::max<long>(1, strtol(e, nullptr, 10));   // test knob: window size (triples kept hot)
                B.hotbits_min = std::min(16u, std::max(10u, ceil_log2((uint32_t)(4u * std::min<uint64_t>(pairs0, 16384u)))));
                B.hotbits_max = std::max(B.hotbits_min, std::min(19u, ceil_log2((uint32_t)(4u * hotmax + 1024u))));
                if (const char *e = getenv("PLO_BIG_HOTBITS")) B.hotbits_min = (uint32_t)std::min<long>(B.hotbits_max, std::max<long>(6, strtol(e, nullptr, 10)));   // test knob: a full hot table is reported and the launch repeated with a larger one
                B.lgrp = 3000u; if (const char *e = getenv("PLO_BIG_LGRP")) B.lgrp = (uint32_t)std::min<long>(7000, std::max<long>(64, strtol(e, nullptr, 10)));   // experiment knob: records per group of partitions summed together
                st0.assign((size_t)capp * Pn, 0ull);
                std::vector<uint32_t> fill(Pn, 0u);
                for (size_t k = 0; k < keys.size(); ++k) { const uint32_t q = part(keys[k]); st0[(size_t)q * capp + fill[q]++] = (keys[k] << PLO_GVB) | 0x8000ull | cnts[k]; }   // a record: key | insert flag | frequency
            }
        }
    }
    int rc;
    if ((rc = upload(pl, pl->rowptr, &B.rs)) || (rc = upload(pl, ent, &B.ent0)) || (rc = upload(pl, vt, &B.vt)) ||
        (rc = upload(pl, tptr, &B.tptr)) || (rc = upload(pl, trows, &B.trows)) ||
        (rc = upload(pl, ucount, &B.ucount0)) || (rc = upload(pl, hist, &B.hist0))) return rc;
    if (B.defer) { if ((rc = upload(pl, st0, &B.st0)) || (rc = upload(pl, pc0, &B.pcount0))) return rc; B.tab0 = nullptr; }
    else {
        std::vector<uint64_t> tab(cap, PLO_GEMPTY);
        for (size_t k = 0; k < keys.size(); ++k) {
            uint32_t s = (uint32_t)((keys[k] * 0x9E3779B97F4A7C15ull) >> (64u - hbits));       // == plo::ghash
            while (tab[s] != PLO_GEMPTY) s = (s + 1) & (uint32_t)(cap - 1);
            tab[s] = (keys[k] << PLO_GVB) | cnts[k];
        }
        if ((rc = upload(pl, tab, &B.tab0))) return rc;
    }
    B.nv = (uint32_t)dv.size(); B.vt_lds = (dv.size() <= 512 && !getenv("PLO_BIG_VT_GLOBAL")) ? 1u : 0u;   // (test knob: the global-memory value table)
    B.mode = B.vt_lds ? 1u : 0u; B.nr = 0;
    if (ratio_ids) {
        const uint32_t nv = (uint32_t)dv.size();
        std::vector<uint16_t> rt(PLO_RSTRIDE * PLO_RSTRIDE, 0), iv(rv.size());     // identifiers at [i * 32 + j]
        for (uint32_t i = 0; i < nv; ++i) for (uint32_t j = 0; j < nv; ++j) rt[i * PLO_RSTRIDE + j] = (uint16_t)(std::lower_bound(rv.begin(), rv.end(), rat[i * nv + j]) - rv.begin());
        for (size_t k = 0; k < rv.size(); ++k) iv[k] = (uint16_t)(std::lower_bound(rv.begin(), rv.end(), inv_mod(rv[k], p)) - rv.begin());   // the inverse of v_i/v_j is v_j/v_i: in the set
        std::vector<uint8_t> ng(PLO_RSTRIDE, 0xFF);                                 // value index of -v
        for (uint32_t i = 0; i < nv; ++i) { const auto it = std::lower_bound(dv.begin(), dv.end(), p - dv[i]); if (it != dv.end() && *it == p - dv[i]) ng[i] = (uint8_t)(it - dv.begin()); }
        if ((rc = upload(pl, rv, &B.rval)) || (rc = upload(pl, rt, &B.rtid)) || (rc = upload(pl, iv, &B.invid)) || (rc = upload(pl, ng, &B.negidx))) return rc;
        B.nr = (uint32_t)rv.size(); B.mode = 2u;
        B.id_one = (uint32_t)(std::lower_bound(rv.begin(), rv.end(), 1u) - rv.begin());
        { const auto it = std::lower_bound(rv.begin(), rv.end(), p - 1u); B.id_mone = (it != rv.end() && *it == p - 1u) ? (uint32_t)(it - rv.begin()) : 0xFFFFu; }
    }
    B.invtab = nullptr;
    if (p <= (1u << 20)) {                                   // 1/x for every residue (the flush needs v_a/v_c from v_c/v_a): i^-1 = -(p/i) (p mod i)^-1
        std::vector<uint32_t> it(p, 0); it[1] = 1;
        for (uint32_t i = 2; i < p; ++i) it[i] = (uint32_t)((uint64_t)(p - p / i) * it[p % i] % p);
        if ((rc = upload(pl, it, &B.invtab))) return rc;
    }
    // workspace layout of one candidate
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) { uint64_t o = off; off = (off + bytes + 255) & ~255ull; return o; };
    if (B.defer) {
        // the table region only serves ProgramGen's (column, |v|) multiset: it shares the partitions' logs, idle by then
        uint64_t pg = 1024; while (pg < 2ull * nnz + 2ull * multcap + 64ull) pg <<= 1;
        B.hbits = ceil_log2((uint32_t)pg);
        B.o_store = take(std::max<uint64_t>(((uint64_t)(B.capp + B.plcap) << B.pbits) * 8, pg * 8)); B.o_tab = B.o_store; B.o_plog = 0;
        B.o_pcount = take(4ull << B.pbits); B.o_ptail = take(4ull << B.pbits);
        B.o_log = take((uint64_t)B.logcap * 8); B.o_hot = take(8ull << B.hotbits_max);
    } else B.o_tab = take(cap * 8);
    B.o_ent = take(((uint64_t)nnz + 128) * 4); B.o_col = take((uint64_t)nnz * 4); B.o_val = take((uint64_t)nnz * 4); B.o_inv = take((uint64_t)nnz * 4);
    B.o_len = take((uint64_t)m * 4); B.o_ucount = take(NC * 4); B.o_cntM = take(NC * 4);
    B.o_dm = take((uint64_t)B.dmcap * 8); B.o_hl = take((uint64_t)B.hlcap * 16); B.o_aff = take((uint64_t)m * 32);
    B.o_ncrptr = take((NC + 2) * 4); B.o_ncr = take(((uint64_t)nnz + 64) * 4);
    B.o_tl = take(((uint64_t)nnz + 64) * 4); B.o_clen = take(NC * 4); B.o_keep = take(((uint64_t)m + 64) * 4);
    B.o_multc = take((uint64_t)multcap * 4); B.o_multv = take((uint64_t)multcap * 4);
    B.o_tcnt = take(NC * 4); B.o_tptr2 = take((NC + 2) * 4); B.o_tlist = take(((uint64_t)nnz + 64) * 4); B.o_cols2 = take(NC * 4); B.o_spill = take(((uint64_t)nnz + 64) * 8);
    B.ws_stride = off;
    // dynamic LDS: histogram + max(ProgramGen scratch, aggregation table of 2^aggbits u64)
    B.aggbits = std::min(13u, std::max(6u, ceil_log2((uint32_t)std::min<uint64_t>(4 * pairs0 + 64, 1u << 13))));
    if (const char *e = getenv("PLO_BIG_AGGBITS")) B.aggbits = (uint32_t)std::min(14l, std::max(6l, strtol(e, nullptr, 10)));
    // LDS aggregation entry: key | count.  When column, ratio, inverse ratio and a count up to m fit 64 bits the key carries
    // both x and 1/x (the flush then needs no inversion); otherwise (column, x) with a 16-bit count.
    { const uint32_t cb = ceil_log2(m + 2u);
      if (bb + 2u * rb + cb <= 64u && !getenv("PLO_BIG_NODUAL")) { B.agg_dual = 1u; B.agg_cb = 64u - bb - 2u * rb; if (B.agg_cb > 16u) B.agg_cb = 16u; }
      else { B.agg_dual = 0u; B.agg_cb = 16u; } }
    // dynamic LDS, in words: histogram, tables of the mode, then max(ProgramGen scratch, aggregation table: 2^aggbits entries of 8 bytes, 6 in mode 2)
    // Mode 2 with deferred updates (the flat sweep): beside the hashed aggregation table the scratch region holds a DIRECT count table
    // indexed by column for the entries whose ratio is +-1 (84 % of config 5's), and the waves' queues of the other entries
    // (plo_cse_big.hip, "direct counts"); the hashed table then has 2^12 slots at most.
#ifdef PLO_BIG_DIRECT
    const bool direct = B.mode == 2u && B.defer && !getenv("PLO_BIG_NODIRECT");
#else
    const bool direct = false;                               // (an experiment of round 4, kept behind -DPLO_BIG_DIRECT: DESIGN.md 2.2)
#endif
    if (direct && B.aggbits > 12u && !getenv("PLO_BIG_AGGBITS")) B.aggbits = 12u;
    const uint32_t agg_words = B.mode == 2u ? (1u << B.aggbits) + (1u << B.aggbits) / 2u : 2u << B.aggbits;
    uint32_t scr_words = std::max<uint32_t>((PLO_BIG_THREADS / 64) * maxlen, agg_words);
    uint32_t tab_words = B.mode == 1u ? 2u * ((B.nv + 1u) & ~1u) : B.mode == 2u ? ((B.nr + 1u) & ~1u) + PLO_RSTRIDE * PLO_RSTRIDE / 2u + (B.nr + 3u) / 4u * 2u + PLO_RSTRIDE / 4u + (B.defer ? 0u : (1u << B.aggbits) / 2u) : 0u;   // mode 2: ratio values, ratio ids, inverse ids, negated value indices, slot list of the aggregation table (eager flush only)
    if (B.defer) { tab_words += PLO_DBLOOM_WORDS; scr_words = std::max<uint32_t>(scr_words, PLO_DMREG_WORDS - PLO_DBLOOM_WORDS); }   // Bloom filter, and 64 KB in all for the merge
    B.dcols = 0u;
    if (direct && scr_words >= agg_words + PLO_BIG_QUEUE_WORDS + 1024u) {
        B.dcols = std::min<uint32_t>(scr_words - agg_words - PLO_BIG_QUEUE_WORDS, 32768u);
        if (const char *e = getenv("PLO_BIG_DCOLS")) B.dcols = (uint32_t)std::min<long>(B.dcols, std::max<long>(1, strtol(e, nullptr, 10)));   // test knob: columns beyond go through the queue
    }
    pl->big_lds = (((B.maxf0 + 2u) & ~1u) + tab_words + scr_words) * 4u;
    if (pl->big_lds + sizeof(plo::BigShared) + 64 > g_lds_max) return fail(PLO_E_CAPACITY, "frequency histogram does not fit LDS");
    HIPCHK(hipFuncSetAttribute(big_kernel_fn(B), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->big_lds));
    if (!pl->d_err) HIPCHK(hipMalloc((void **)&pl->d_err, sizeof(uint32_t)));
    if (!pl->d_best) HIPCHK(hipMalloc((void **)&pl->d_best, sizeof(unsigned long long)));
    if (!pl->d_next) HIPCHK(hipMalloc((void **)&pl->d_next, sizeof(unsigned long long)));
    if (!pl->d_stats) HIPCHK(hipMalloc((void **)&pl->d_stats, 64 * sizeof(uint32_t)));
    B.selcap = PLO_BIG_SELCAP;
    if (const char *e = getenv("PLO_BIG_SELCAP")) B.selcap = (uint32_t)std::min<long>(PLO_BIG_SELCAP, std::max<long>(1, strtol(e, nullptr, 10)));   // test knob: forces the bisection tie pick
    pl->big = true; pl->waves_per_wg = PLO_BIG_THREADS / 64; pl->lds_bytes = pl->big_lds + (uint32_t)sizeof(plo::BigShared);
    return PLO_OK;
}

// launch of the HBM-resident kernel over J.ncand candidates
int launch_big(plo_plan *pl, plo::BigJob J, plo_stats_t *st)
{
    // one workspace slice per resident workgroup
    uint64_t per_cu = 2;                  // = the resident workgroups (LDS-limited); more slices only enlarge the footprint
    if (const char *e = getenv("PLO_BIG_WG_PER_CU")) per_cu = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    uint64_t want = std::min<uint64_t>(J.ncand, (uint64_t)g_cus * per_cu);
    if (const char *e = getenv("PLO_BIG_SLICES")) want = std::min<uint64_t>(want, strtoull(e, nullptr, 10));
    if (want == 0) want = 1;
    if (pl->ws_slices < want) {
        if (pl->d_ws) { (void)hipFree(pl->d_ws); pl->d_ws = nullptr; pl->ws_slices = 0; }
        size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot));
        uint64_t fit = (uint64_t)(fr * 0.85) / pl->B.ws_stride;
        if (fit == 0) return fail(PLO_E_CAPACITY, "not enough HBM for one candidate workspace");
        want = std::min(want, fit);
        HIPCHK(hipMalloc(&pl->d_ws, want * pl->B.ws_stride));
        pl->ws_slices = want;
    }
    const uint64_t grid = std::min<uint64_t>(pl->ws_slices, want);
    pl->B.ws = (uint8_t *)pl->d_ws;
    HIPCHK(hipMemsetAsync(pl->d_err, 0, sizeof(uint32_t), g_stream));
    HIPCHK(hipMemsetAsync(pl->d_next, 0, sizeof(unsigned long long), g_stream));
    HIPCHK(hipMemsetAsync(pl->d_stats, 0, 64 * sizeof(uint32_t), g_stream));
    J.err = pl->d_err; J.next = pl->d_next; J.stats = pl->d_stats;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g_stream));
    {
        const plo::BigPlan &B = pl->B;
        if (B.defer) {
            if (B.idk) hipLaunchKernelGGL((plo::cse_big_kernel<2, true, true>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else if (B.mode == 2u) hipLaunchKernelGGL((plo::cse_big_kernel<2, true>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else if (B.mode == 1u) hipLaunchKernelGGL((plo::cse_big_kernel<1, true>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else hipLaunchKernelGGL((plo::cse_big_kernel<0, true>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
        } else {
            if (B.idk) hipLaunchKernelGGL((plo::cse_big_kernel<2, false, true>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else if (B.mode == 2u) hipLaunchKernelGGL((plo::cse_big_kernel<2, false>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else if (B.mode == 1u) hipLaunchKernelGGL((plo::cse_big_kernel<1, false>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
            else hipLaunchKernelGGL((plo::cse_big_kernel<0, false>), dim3((uint32_t)grid), dim3(PLO_BIG_THREADS), pl->big_lds, g_stream, pl->B, J);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, g_stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, pl->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (st) { st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = pl->lds_bytes; st->waves_per_wg = pl->waves_per_wg; st->algo_bytes = pl->algo_bytes; }
    if (getenv("PLO_BIG_STATS")) {
        uint32_t hs[64] = {0};
        if (hipMemcpy(hs, pl->d_stats, sizeof hs, hipMemcpyDeviceToHost) == hipSuccess) {
            if (pl->B.defer && hs[38]) fprintf(stderr, "# big kernel, deferred updates: per candidate %.1f merges (%.2f forced by log/hot pressure), %.0f log records, %.0f hot-table updates; last candidate, merge us: hot->log %u, partition pass %u, sum + write back %u, window %u\n",
                    (double)hs[33] / hs[38], (double)hs[39] / hs[38], ((double)hs[42] * 4294967296.0 + hs[41]) / hs[38], (double)hs[40] / hs[38], hs[44], hs[45], hs[46], hs[47]);
            if (pl->B.defer && hs[52]) fprintf(stderr, "#   merge, sum + write back of the last candidate: %u groups; us: sum (loads + table) %u, scan + write back %u, clear + bounds %u\n", hs[52], hs[48], hs[49], hs[51]);
            fprintf(stderr, "# big kernel (last candidate): steps %u, full scans %u, level rebuilds %u; phase us: level %u select %u rows %u sweep1 %u flush1 %u sweep2 %u flush2 %u tail %u\n",
                    hs[0], hs[1], hs[2], hs[4], hs[5], hs[6], hs[7], hs[8], hs[9], hs[10], hs[11]);
            fprintf(stderr, "# big kernel (last candidate): image load + CSE phase %u us, ProgramGen %u us\n", hs[55], hs[56]);
#ifdef PLO_BIG_PROFILE
            { unsigned long long g2[40] = {0}; if (hipMemcpyFromSymbol(g2, HIP_SYMBOL(plo::g_prof2), sizeof g2) == hipSuccess && g2[32]) {
                static const char *cls[4] = {">=256", "64..255", "16..63", "<16"};
                for (int c_ = 0; c_ < 4; ++c_) { fprintf(stderr, "#   rows/step %-8s ms per candidate: level %.1f select %.1f rows %.1f sweep %.1f flush1 %.1f flush2 %.1f tail %.1f\n", cls[c_],
                    g2[c_ * 8 + 0] / 1e5 / g2[32], g2[c_ * 8 + 1] / 1e5 / g2[32], g2[c_ * 8 + 2] / 1e5 / g2[32], g2[c_ * 8 + 3] / 1e5 / g2[32], g2[c_ * 8 + 4] / 1e5 / g2[32], (g2[c_ * 8 + 6] + g2[c_ * 8 + 5]) / 1e5 / g2[32], g2[c_ * 8 + 7] / 1e5 / g2[32]); } } }
            { unsigned long long gp[16] = {0}; if (hipMemcpyFromSymbol(gp, HIP_SYMBOL(plo::g_prof), sizeof gp) == hipSuccess && gp[3]) fprintf(stderr, "#   sweep of the steps with >= 256 rows, all candidates: %llu trips by %llu wave-sweeps; cycles per trip: chunk wait + stores %.0f, aggregation of both chunks %.0f, rest of the loop %.0f; per wave-sweep %.0f cycles, %.1f trips; probe rounds per trip %.2f, active lanes per trip %.1f\n", gp[3], gp[5], (double)gp[0] / gp[3], (double)gp[1] / gp[3], (double)gp[2] / gp[3], (double)gp[4] / gp[5], (double)gp[3] / gp[5], (double)gp[6] / gp[3], (double)gp[7] / gp[3]); }
            { unsigned long long gp[16] = {0}; if (hipMemcpyFromSymbol(gp, HIP_SYMBOL(plo::g_prof), sizeof gp) == hipSuccess && gp[3] && pl->B.defer && pl->B.mode == 2u) fprintf(stderr, "#   aggregation of a trip (lane 0 of every wave, big steps), cycles: ratio-id lookup %.0f, pair read %.0f, compare-and-swap %.0f, bitmap + count %.0f\n", (double)gp[14] / gp[3], (double)gp[8] / gp[3], (double)gp[9] / gp[3], (double)gp[10] / gp[3]); }
            { unsigned long long gp[16] = {0}; if (hipMemcpyFromSymbol(gp, HIP_SYMBOL(plo::g_prof), sizeof gp) == hipSuccess && (gp[14] || gp[15]) && !pl->B.defer) fprintf(stderr, "#   flush 1, all candidates: entries whose pair with a has a as SECOND column %llu, with b %llu\n", gp[14], gp[15]); }
            { unsigned long long gp[16] = {0}; if (hipMemcpyFromSymbol(gp, HIP_SYMBOL(plo::g_prof), sizeof gp) == hipSuccess && gp[11]) fprintf(stderr, "#   flush 1 of the steps with >= 256 rows: %llu wave-trips by %llu wave-flushes; cycles per trip: fetch + decode %.0f, probe loads %.0f, stores + bookkeeping %.0f; per wave-flush %.0f cycles, %.1f trips\n", gp[11], gp[13], (double)gp[8] / gp[11], (double)gp[9] / gp[11], (double)gp[10] / gp[11], (double)gp[12] / gp[13], (double)gp[11] / gp[13]); }
            fprintf(stderr, "#   steps by rows/step [>=256, 64.., 16.., <16]: %u %u %u %u; sweep1 us %u %u %u %u; sweep2 us %u %u %u %u; fallbacks %u %u; flushed keys %u %u\n",
                    hs[24], hs[25], hs[26], hs[27], hs[16], hs[17], hs[18], hs[19], hs[20], hs[21], hs[22], hs[23], hs[28], hs[29], hs[30], hs[31]);
#endif
        }
    }
    if (err == plo::BERR_TABLE && !pl->B.defer && pl->big_cap_scale < 256u) {
        // eager table full: the live triples of frequency >= 2 can outnumber the input's (new columns pair with every column of the rows
        // they enter); four times the slots and again (the workspace grows, fewer candidates are resident)
        pl->big_cap_scale *= 4u; ++pl->big_refits;
        for (void *d : pl->big_bufs) (void)hipFree(d);
        pl->big_bufs.clear();
        if (pl->d_ws) { (void)hipFree(pl->d_ws); pl->d_ws = nullptr; pl->ws_slices = 0; }
        const int rc = build_big_plan(pl);
        if (rc != PLO_OK) return rc;
        return launch_big(pl, J, st);
    }
    if (err == plo::BERR_TABLE && pl->B.defer && !pl->big_no_defer) {
        // The structures of the deferred updates are sized from the INPUT's triples (a partition's live triples: its initial share + 25 %;
        // the log and the hot table from the longest rows); a candidate whose live triples grow beyond that -- random dense matrices
        // with few distinct values do (tests/soak_hbm.py) -- is reported by the device.  The plan is rebuilt with the eager table of
        // round 2 (sized for every pair instance of the input, dead slots reclaimed) and the launch repeated: same results, slower.
        pl->big_no_defer = true; ++pl->big_refits;
        for (void *d : pl->big_bufs) (void)hipFree(d);
        pl->big_bufs.clear();
        if (pl->d_ws) { (void)hipFree(pl->d_ws); pl->d_ws = nullptr; pl->ws_slices = 0; }
        const int rc = build_big_plan(pl);
        if (rc != PLO_OK) return rc;
        return launch_big(pl, J, st);
    }
    if (err) {
        static const char *names[] = {"pair table", "frequency/row-count mismatch", "column bound", "level list", "window list", "multiplier list", "tie selection", "ProgramGen"};
        uint32_t site = 0; (void)hipMemcpy(&site, pl->d_stats + 43, sizeof site, hipMemcpyDeviceToHost);
        // BERR_PGEN: ProgramGen's Triangle keeps the rows of a column one per lane (more than 64 rows with a non +-1 entry in one column), or its
        // multiset does not fit the table region: a limit of this build, not an inconsistency -- the tools then search on the host
        return fail(err == plo::BERR_COLS || err == plo::BERR_DM || err == plo::BERR_HL ? PLO_E_CAPACITY : err == plo::BERR_PGEN ? PLO_E_UNSUPPORTED : PLO_E_INTERNAL,
                    std::string("device (HBM variant): ") + (err >= 11 && err <= 18 ? names[err - 11] : "unknown") + " error " + std::to_string(err) + (site ? " (site " + std::to_string(site) + ")" : ""));
    }
    return PLO_OK;
}

// one launch over [first, first+count) candidates of a job
int launch(plo_plan *pl, plo::WaveJob J, plo_stats_t *st, float *ms_out)
{
    const uint32_t W = pl->waves_per_wg;
    uint64_t need = (J.ncand + W - 1) / W;
    uint64_t grid = std::min<uint64_t>((uint64_t)g_cus * pl->blocks_per_cu, need);
    if (grid == 0) grid = 1;
    HIPCHK(hipMemsetAsync(pl->d_err, 0, sizeof(uint32_t), g_stream));
    J.err = pl->d_err;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g_stream));
    if (pl->P.unit) hipLaunchKernelGGL(plo::cse_wave_kernel<true>, dim3((uint32_t)grid), dim3(W * 64), pl->lds_bytes, g_stream, pl->P, J);
    else hipLaunchKernelGGL(plo::cse_wave_kernel<false>, dim3((uint32_t)grid), dim3(W * 64), pl->lds_bytes, g_stream, pl->P, J);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, g_stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, pl->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (ms_out) *ms_out = ms;
    if (st) { st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = pl->lds_bytes; st->waves_per_wg = W; st->algo_bytes = pl->algo_bytes; }
#ifdef PLO_WAVE_PROFILE
    if (getenv("PLO_WAVE_STATS")) {
        unsigned long long g[12] = {0};
        if (hipMemcpyFromSymbol(g, HIP_SYMBOL(plo::g_wprof), sizeof g) == hipSuccess && g[9])
            fprintf(stderr, "# wave kernel, cumulative: %llu candidates, %.1f steps each; cycles per step: max scan %.0f, tie list %.0f, tie pick %.0f, sweep 1 %.0f, sweep 2 %.0f, tail %.0f; sweep 1 trips per step %.2f, per trip: loads + ballots + broadcasts %.0f, retirements %.0f\n",
                    g[9], (double)g[8] / g[9], (double)g[0] / g[8], (double)g[1] / g[8], (double)g[2] / g[8], (double)g[3] / g[8], (double)g[4] / g[8], (double)g[5] / g[8], (double)g[11] / g[8], g[11] ? (double)g[6] / g[11] : 0.0, g[11] ? (double)g[7] / g[11] : 0.0);
    }
#endif
    return (int)err;   // >0: device error word
}

int device_error(int err) {
    switch (err) {
    case plo::ERR_MULT:  return fail(PLO_E_INTERNAL, "device: multiplier list overflow");
    case plo::ERR_STEPS: return fail(PLO_E_INTERNAL, "device: more CSE steps than the column bound");
    case plo::ERR_PGEN:  return fail(PLO_E_UNSUPPORTED, "device: ProgramGen for non +-1 coefficients not available in this build");
    case plo::ERR_KDEC:  return fail(PLO_E_INTERNAL, "device: nullspace decomposition inconsistent with the rank found on the host");
    default:             return fail(PLO_E_INTERNAL, "device: pair table inconsistency");
    }
}

// run a job, growing the pair table when the device reports it full
int run_job(plo_plan *pl, plo::WaveJob J, plo_stats_t *st)
{
    for (int attempt = 0; attempt < 4; ++attempt) {
        int err = launch(pl, J, st, nullptr);
        if (err < 0) return err;
        if (err == 0) return PLO_OK;
        if (err != plo::ERR_TABLE) return device_error(err);
        pl->cap_scale *= 2;                       // table full: re-plan with twice the slots and retry
        int rc = build_plan(pl);
        if (rc != PLO_OK) return rc;
        if (J.best) HIPCHK(hipMemsetAsync(pl->d_best, 0xFF, sizeof(unsigned long long), g_stream));
    }
    return device_error(plo::ERR_TABLE);
}


} // namespace

// Two prepared matrices evaluated back to back per candidate with one random stream (LU method)
struct plo_chain {
    plo_plan *st[2] = {nullptr, nullptr};
    uint32_t W = 1, lds = 0, blocks_per_cu = 1;
};

namespace {

int chain_config(plo_chain *ch)
{
    const plo::WavePlan &A = ch->st[0]->P, &B = ch->st[1]->P;
    const uint32_t region = std::max(A.region_bytes, B.region_bytes), fixed = A.rs_bytes + B.rs_bytes;
    if (fixed + region > g_lds_max) return fail(PLO_E_CAPACITY, "chained candidate state does not fit LDS");
    uint32_t W = 1, bestw = 0;
    for (uint32_t w : {4u, 2u, 1u}) {
        const uint32_t lds = fixed + w * region;
        if (lds > g_lds_max) continue;
        const uint32_t waves = std::min<uint32_t>(32u, (uint32_t)(g_lds_max / lds) * w);
        if (waves > bestw) { bestw = waves; W = w; }
    }
    ch->W = W; ch->lds = fixed + W * region;
    HIPCHK(hipFuncSetAttribute((const void *)plo::cse_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ch->lds));
    int nb = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)plo::cse_chain_kernel, (int)(W * 64), ch->lds));
    ch->blocks_per_cu = (uint32_t)std::max(nb, 1);
    return PLO_OK;
}

int run_chain(plo_chain *ch, plo::WaveJob J, plo_stats_t *st)
{
    plo_plan *p0 = ch->st[0];
    for (int attempt = 0; attempt < 4; ++attempt) {
        const uint64_t need = (J.ncand + ch->W - 1) / ch->W;
        const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g_cus * ch->blocks_per_cu, need));
        HIPCHK(hipMemsetAsync(p0->d_err, 0, sizeof(uint32_t), g_stream));
        J.err = p0->d_err;
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, g_stream));
        hipLaunchKernelGGL(plo::cse_chain_kernel, dim3((uint32_t)grid), dim3(ch->W * 64), ch->lds, g_stream, ch->st[0]->P, ch->st[1]->P, J);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, g_stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        uint32_t err = 0;
        HIPCHK(hipMemcpy(&err, p0->d_err, sizeof err, hipMemcpyDeviceToHost));
        if (st) { st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = ch->lds; st->waves_per_wg = ch->W;
                  st->algo_bytes = ch->st[0]->algo_bytes + ch->st[1]->algo_bytes; }
        if (err == 0) return PLO_OK;
        if (err != plo::ERR_TABLE) return device_error((int)err);
        for (int k = 0; k < 2; ++k) { ch->st[k]->cap_scale *= 2; int rc = build_plan(ch->st[k]); if (rc != PLO_OK) return rc; }
        int rc = chain_config(ch); if (rc != PLO_OK) return rc;
        if (J.best) HIPCHK(hipMemsetAsync(p0->d_best, 0xFF, sizeof(unsigned long long), g_stream));
    }
    return device_error(plo::ERR_TABLE);
}

} // namespace

extern "C" {

const char *plo_last_error(void) { return g_err.c_str(); }

int plo_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int plo_init(int device)
{
    int n = 0;
    hipError_t ce = hipGetDeviceCount(&n);
    if (ce != hipSuccess || n <= 0)
        return fail(PLO_E_HIP, std::string("no HIP device visible (") + hipGetErrorString(ce) + "): libplinopt_hip has no CPU fallback");
    if (device < 0 || device >= n) return fail(PLO_E_ARG, "device ordinal out of range");
    if (g_device == device && g_stream) { HIPCHK(hipSetDevice(device)); return PLO_OK; }      // (the calling thread may have been moved to another device since)
    if (g_device >= 0 && g_device != device) plo_shutdown();          // scratch and stream of the device left behind
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    g_cus = prop.multiProcessorCount;
    g_lds_max = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    if (g_lds_max > 160u * 1024u) g_lds_max = 160u * 1024u;
    if (g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
    HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
    return PLO_OK;
}

int plo_shutdown(void)
{
    DevCtx &cx = cur_ctx();
    if (cx.cob_buf) { (void)hipFree(cx.cob_buf); cx.cob_buf = nullptr; cx.cob_words = 0; }
    if (cx.cob_e0) { (void)hipEventDestroy(cx.cob_e0); (void)hipEventDestroy(cx.cob_e1); cx.cob_e0 = cx.cob_e1 = nullptr; }
    if (g_stream) { (void)hipStreamSynchronize(g_stream); (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
    g_device = -1;
    return PLO_OK;
}

int plo_cse_plan_create_ex(const plo_csr_t *A, uint32_t p, uint32_t flags, plo_plan_t **out)
{
    if (!A || !out || !A->rowptr || (A->rowptr[A->m] && (!A->col || !A->val))) return fail(PLO_E_ARG, "null argument");
    if (p < 3 || p >= 0x80000000u || !(p & 1u)) return fail(PLO_E_ARG, "modulus must be an odd prime below 2^31");
    if (g_device < 0) { int rc = plo_init(0); if (rc != PLO_OK) return rc; }
    if (const char *bad = csr_defect(A, p)) return fail(PLO_E_ARG, bad);
    plo_plan *pl = new plo_plan();
    pl->m = A->m; pl->n = A->n; pl->p = p;
    pl->rowptr.assign(A->rowptr, A->rowptr + A->m + 1);
    pl->col.assign(A->col, A->col + A->rowptr[A->m]);
    pl->val.assign(A->val, A->val + A->rowptr[A->m]);
    int rc = (flags & PLO_PLAN_HBM) ? PLO_E_CAPACITY : build_plan(pl);
    if (rc == PLO_E_CAPACITY) rc = build_big_plan(pl);       // does not fit LDS: HBM-resident kernel family
    if (rc != PLO_OK) { plo_cse_plan_destroy(pl); return rc; }
    *out = pl;
    return PLO_OK;
}

int plo_cse_plan_create(const plo_csr_t *A, uint32_t p, plo_plan_t **out) { return plo_cse_plan_create_ex(A, p, 0u, out); }

int plo_cse_plan_is_hbm(const plo_plan_t *pl) { return pl && pl->big ? 1 : 0; }

int plo_cse_plan_hbm_counters(const plo_plan_t *pl, uint32_t out[8])
{
    if (!pl || !out) return fail(PLO_E_ARG, "null argument");
    if (!pl->big || !pl->d_stats) return fail(PLO_E_UNSUPPORTED, "the counters belong to the HBM-resident kernel family");
    uint32_t hs[64];
    HIPCHK(hipMemcpy(hs, pl->d_stats, sizeof hs, hipMemcpyDeviceToHost));
    for (int k = 0; k < 7; ++k) out[k] = hs[32 + k];
    out[7] = pl->big_refits;
    return PLO_OK;
}

int plo_cse_plan_hbm_counters_ex(const plo_plan_t *pl, uint32_t *out, uint32_t n)
{
    if (!pl || !out || n > 10u) return fail(PLO_E_ARG, "bad argument");
    uint32_t o[10] = {0};
    const int rc = plo_cse_plan_hbm_counters(pl, o);
    if (rc != PLO_OK) return rc;
    uint32_t hs[64];
    HIPCHK(hipMemcpy(hs, pl->d_stats, sizeof hs, hipMemcpyDeviceToHost));
    o[8] = hs[53]; o[9] = hs[54];
    for (uint32_t k = 0; k < n; ++k) out[k] = o[k];
    return PLO_OK;
}

int plo_cse_plan_destroy(plo_plan_t *pl)
{
    if (!pl) return PLO_OK;
    if (pl->d_tmpl) (void)hipFree(pl->d_tmpl);
    if (pl->d_err) (void)hipFree(pl->d_err);
    if (pl->d_best) (void)hipFree(pl->d_best);
    for (void *d : pl->big_bufs) (void)hipFree(d);
    if (pl->d_ws) (void)hipFree(pl->d_ws);
    if (pl->d_next) (void)hipFree(pl->d_next);
    if (pl->d_stats) (void)hipFree(pl->d_stats);
    delete pl;
    return PLO_OK;
}

uint64_t plo_pack_cost(uint32_t adds, uint32_t muls, int cost_mode, uint32_t seed_off)
{
    uint32_t key;
    switch (cost_mode) {
    case PLO_COST_ADD_THEN_MUL: key = (adds << 16) | muls; break;
    case PLO_COST_SUM:          key = (adds + muls) << 16; break;
    default:                    key = ((adds + muls) << 16) | adds; break;
    }
    return ((uint64_t)key << 32) | seed_off;
}

int plo_cse_cost_many_plan(plo_plan_t *pl, const uint64_t *seeds, uint64_t seed0, uint64_t n,
                           uint32_t *adds, uint32_t *muls, plo_stats_t *st)
{
    if (!pl || !adds || !muls) return fail(PLO_E_ARG, "null argument");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    if (n == 0) return PLO_OK;
    uint32_t *d_adds = nullptr, *d_muls = nullptr; uint64_t *d_seeds = nullptr;
    HIPCHK(hipMalloc((void **)&d_adds, n * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&d_muls, n * sizeof(uint32_t)));
    if (seeds) { HIPCHK(hipMalloc((void **)&d_seeds, n * sizeof(uint64_t))); HIPCHK(hipMemcpy(d_seeds, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice)); }
    int rc;
    if (pl->big) {
        plo::BigJob J{}; J.seed0 = seed0; J.seeds = d_seeds; J.ncand = n; J.adds = d_adds; J.muls = d_muls; J.best = nullptr; J.cost_mode = 0;
        rc = launch_big(pl, J, st);
    } else {
        plo::WaveJob J{}; J.seed0 = seed0; J.seeds = d_seeds; J.ncand = n; J.adds = d_adds; J.muls = d_muls; J.best = nullptr; J.cost_mode = 0;
        rc = run_job(pl, J, st);
    }
    if (rc == PLO_OK) {
        hipError_t e1 = hipMemcpy(adds, d_adds, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(muls, d_muls, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(PLO_E_HIP, "copy back failed");
    }
    (void)hipFree(d_adds); (void)hipFree(d_muls); if (d_seeds) (void)hipFree(d_seeds);
    st->candidates = n;
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int plo_cse_cost_many(const plo_csr_t *A, uint32_t p, const uint64_t *seeds, uint64_t seed0, uint64_t n,
                      uint32_t *adds, uint32_t *muls)
{
    plo_plan_t *pl = nullptr;
    int rc = plo_cse_plan_create(A, p, &pl);
    if (rc != PLO_OK) return rc;
    rc = plo_cse_cost_many_plan(pl, seeds, seed0, n, adds, muls, nullptr);
    plo_cse_plan_destroy(pl);
    return rc;
}

int plo_cse_search_plan(plo_plan_t *pl, uint64_t seed0, uint64_t nseeds, int cost_mode,
                        plo_best_t *out, plo_stats_t *st)
{
    if (!pl || !out) return fail(PLO_E_ARG, "null argument");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    out->adds = out->muls = 0xFFFFFFFFu; out->seed = ~0ull;
    uint64_t bkey = ~0ull, bseed = ~0ull;
    if (pl->big) {
        const uint64_t CHB = 1ull << 24;                      // 24-bit seed offsets in the 64-bit cost word of the HBM variant
        uint32_t ba = 0, bm = 0;
        for (uint64_t done = 0; done < nseeds;) {
            const uint64_t cnt = std::min<uint64_t>(CHB, nseeds - done);
            HIPCHK(hipMemsetAsync(pl->d_best, 0xFF, sizeof(unsigned long long), g_stream));
            plo::BigJob J{}; J.seed0 = seed0 + done; J.seeds = nullptr; J.ncand = cnt; J.best = pl->d_best; J.cost_mode = (uint32_t)cost_mode;
            int rc = launch_big(pl, J, st);
            if (rc != PLO_OK) return rc;
            unsigned long long w = 0;
            HIPCHK(hipMemcpy(&w, pl->d_best, sizeof w, hipMemcpyDeviceToHost));
            const uint64_t key = w >> 24, sd = seed0 + done + (w & 0xFFFFFFull);
            if (key < bkey || (key == bkey && sd < bseed)) { bkey = key; bseed = sd; }
            done += cnt;
        }
        if (nseeds) {
            if (cost_mode == PLO_COST_SUM_THEN_ADD) { ba = (uint32_t)(bkey & 0xFFFFFu); bm = (uint32_t)(bkey >> 20) - ba; }
            else if (cost_mode == PLO_COST_ADD_THEN_MUL) { ba = (uint32_t)(bkey >> 20); bm = (uint32_t)(bkey & 0xFFFFFu); }
            else { plo_stats_t s2{}; int rc = plo_cse_cost_many_plan(pl, &bseed, 0, 1, &ba, &bm, &s2); if (rc != PLO_OK) return rc; }
            out->adds = ba; out->muls = bm; out->seed = bseed;
        }
        st->candidates = nseeds;
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return PLO_OK;
    }
    const uint64_t CH = 0xFFFFFFFFull;                        // seed offsets inside a launch are 32-bit
    for (uint64_t done = 0; done < nseeds;) {
        const uint64_t cnt = std::min<uint64_t>(CH, nseeds - done);
        HIPCHK(hipMemsetAsync(pl->d_best, 0xFF, sizeof(unsigned long long), g_stream));
        plo::WaveJob J{}; J.seed0 = seed0 + done; J.seeds = nullptr; J.ncand = cnt; J.best = pl->d_best; J.cost_mode = (uint32_t)cost_mode;
        int rc = run_job(pl, J, st);
        if (rc != PLO_OK) return rc;
        unsigned long long w = 0;
        HIPCHK(hipMemcpy(&w, pl->d_best, sizeof w, hipMemcpyDeviceToHost));
        const uint64_t key = w >> 32, sd = seed0 + done + (w & 0xFFFFFFFFull);
        if (key < bkey || (key == bkey && sd < bseed)) { bkey = key; bseed = sd; }
        done += cnt;
    }
    if (nseeds) {
        uint32_t a = 0, mu = 0;
        if (cost_mode == PLO_COST_SUM_THEN_ADD) { a = (uint32_t)(bkey & 0xFFFFu); mu = (uint32_t)(bkey >> 16) - a; }
        else if (cost_mode == PLO_COST_ADD_THEN_MUL) { a = (uint32_t)(bkey >> 16); mu = (uint32_t)(bkey & 0xFFFFu); }
        else {
            plo_stats_t s2{};                                                // sum-only key: ask the device for the split
            int rc = plo_cse_cost_many_plan(pl, &bseed, 0, 1, &a, &mu, &s2);
            if (rc != PLO_OK) return rc;
            if ((plo_pack_cost(a, mu, cost_mode, 0) >> 32) != bkey) return fail(PLO_E_INTERNAL, "winner cost does not match the reduced key");
        }
        out->adds = a; out->muls = mu; out->seed = bseed;
    }
    st->candidates = nseeds;
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PLO_OK;
}

// -E: the schedule space of RecSub walked by index (PickState in plo_cse_wave.hip).  LDS-resident plans only.
int plo_cse_enum_cost_many_plan(plo_plan_t *pl, uint64_t first, uint64_t n, uint32_t *adds, uint32_t *muls, uint64_t *prods, plo_stats_t *st)
{
    if (!pl || !adds || !muls) return fail(PLO_E_ARG, "null argument");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    if (pl->big) return fail(PLO_E_UNSUPPORTED, "schedule enumeration is for matrices of the LDS-resident kernel (the tree of anything larger cannot be walked)");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    if (n == 0) return PLO_OK;
    if (n > 0xFFFFFFFFull) return fail(PLO_E_ARG, "at most 2^32-1 schedules per call");
    uint32_t *d_adds = nullptr, *d_muls = nullptr; unsigned long long *d_prods = nullptr;
    HIPCHK(hipMalloc((void **)&d_adds, n * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&d_muls, n * sizeof(uint32_t)));
    if (prods) HIPCHK(hipMalloc((void **)&d_prods, n * sizeof(unsigned long long)));
    plo::WaveJob J{}; J.seed0 = first; J.seeds = nullptr; J.ncand = n; J.adds = d_adds; J.muls = d_muls; J.best = nullptr; J.cost_mode = 0;
    J.enumerate = 1u; J.prods = d_prods; J.prodmax = nullptr;
    int rc = run_job(pl, J, st);
    if (rc == PLO_OK) {
        hipError_t e1 = hipMemcpy(adds, d_adds, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(muls, d_muls, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        hipError_t e3 = prods ? hipMemcpy(prods, d_prods, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) : hipSuccess;
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(PLO_E_HIP, "copy back failed");
    }
    (void)hipFree(d_adds); (void)hipFree(d_muls); if (d_prods) (void)hipFree(d_prods);
    st->candidates = n;
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int plo_cse_enum_search_plan(plo_plan_t *pl, uint64_t first, uint64_t count, int cost_mode, plo_best_t *out, uint64_t *maxprod, plo_stats_t *st)
{
    if (!pl || !out || !maxprod) return fail(PLO_E_ARG, "null argument");
    if (cost_mode != PLO_COST_SUM_THEN_ADD && cost_mode != PLO_COST_ADD_THEN_MUL && cost_mode != PLO_COST_RECSUB) return fail(PLO_E_ARG, "cost mode 0, 1 or 3");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    if (pl->big) return fail(PLO_E_UNSUPPORTED, "schedule enumeration is for matrices of the LDS-resident kernel (the tree of anything larger cannot be walked)");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    out->adds = out->muls = 0xFFFFFFFFu; out->seed = ~0ull; *maxprod = 0;
    unsigned long long *d_pm = nullptr;
    HIPCHK(hipMalloc((void **)&d_pm, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d_pm, 0, sizeof(unsigned long long), g_stream));
    uint64_t bkey = ~0ull, bidx = ~0ull;
    const uint64_t CH = 0xFFFFFFFFull;
    int rc = PLO_OK;
    for (uint64_t done = 0; done < count && rc == PLO_OK;) {
        const uint64_t cnt = std::min<uint64_t>(CH, count - done);
        if (hipMemsetAsync(pl->d_best, 0xFF, sizeof(unsigned long long), g_stream) != hipSuccess) { rc = fail(PLO_E_HIP, "memset"); break; }
        plo::WaveJob J{}; J.seed0 = first + done; J.seeds = nullptr; J.ncand = cnt; J.best = pl->d_best; J.cost_mode = (uint32_t)cost_mode;
        J.enumerate = 1u; J.prodmax = d_pm;
        rc = run_job(pl, J, st);
        if (rc != PLO_OK) break;
        unsigned long long w = 0;
        if (hipMemcpy(&w, pl->d_best, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(PLO_E_HIP, "copy back"); break; }
        const uint64_t key = w >> 32, ix = first + done + (w & 0xFFFFFFFFull);
        if (key < bkey || (key == bkey && ix < bidx)) { bkey = key; bidx = ix; }
        done += cnt;
    }
    unsigned long long pm = 0;
    if (rc == PLO_OK && hipMemcpy(&pm, d_pm, sizeof pm, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(PLO_E_HIP, "copy back");
    (void)hipFree(d_pm);
    if (rc != PLO_OK) return rc;
    if (count) {
        uint32_t a, mu;
        if (cost_mode == PLO_COST_SUM_THEN_ADD) { a = (uint32_t)(bkey & 0xFFFFu); mu = (uint32_t)(bkey >> 16) - a; }
        else { a = (uint32_t)(bkey >> 16); mu = (uint32_t)(bkey & 0xFFFFu); }
        out->adds = a; out->muls = mu; out->seed = bidx;
    }
    *maxprod = pm;
    st->candidates = count;
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PLO_OK;
}

int plo_cse_search(const plo_csr_t *A, uint32_t p, uint64_t seed0, uint64_t nseeds, int cost_mode,
                   plo_best_t *out, plo_stats_t *stats)
{
    plo_plan_t *pl = nullptr;
    int rc = plo_cse_plan_create(A, p, &pl);
    if (rc != PLO_OK) return rc;
    rc = plo_cse_search_plan(pl, seed0, nseeds, cost_mode, out, stats);
    plo_cse_plan_destroy(pl);
    return rc;
}

} // extern "C" (the helpers below are C++)
// ---- one process, N devices -------------------------------------------------------------------------------------------
// The restart range in `ndev` contiguous shards, one host thread and one device per shard (thread-local device contexts); the
// minimum under (cost key, seed) by RCCL MIN all-reduces over a communicator of the devices.  Shared by plo_cse_search_multi,
// plo_kernel_search_multi and plo_tril_search_multi (below, behind the single-device entries they call).
namespace {
// RCCL, loaded at run time (librccl.so.1 of the ROCm installation; no link-time dependency).
struct Rccl {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    bool ok = false;
    Rccl() {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll"); AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy"); GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        ok = CommInitAll && AllReduce && CommDestroy && GroupStart && GroupEnd;
    }
};
Rccl &rccl() { static Rccl r; return r; }
enum { RCCL_UINT64 = 5, RCCL_MIN = 3 };      // ncclUint64, ncclMin of rccl/rccl.h

// The calling thread's current device is put back when a multi-device entry returns (the all-reduce walks the devices with
// hipSetDevice on the caller's thread; the process-wide context of plo_init still names the caller's device and stream).
struct DeviceGuard { int dev = -1; DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; } ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); } };

// One communicator per device set for the life of the process (ncclCommInitAll over 8 GPUs takes seconds: not per call), with a
// stream and a 16-byte buffer per device.  Never destroyed: the HIP runtime may be gone when static destructors run.
struct CommSet { std::vector<Rccl::comm_t> comms; std::vector<hipStream_t> strm; std::vector<unsigned long long *> dbuf; double init_seconds = 0; };
std::mutex g_comm_mu;
std::map<std::vector<int>, CommSet> &comm_cache() { static auto *m = new std::map<std::vector<int>, CommSet>(); return *m; }
uint64_t g_comm_inits = 0;                    // communicators built so far (tests: a second call on the same devices builds none)

CommSet *comm_for(const std::vector<int> &devs, std::string &why)
{
    auto &cache = comm_cache();
    auto it = cache.find(devs);
    if (it != cache.end()) return &it->second;
    Rccl &R = rccl();
    const int n = (int)devs.size();
    CommSet cs; cs.comms.assign((size_t)n, nullptr); cs.strm.assign((size_t)n, nullptr); cs.dbuf.assign((size_t)n, nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    bool good = R.CommInitAll(cs.comms.data(), n, devs.data()) == 0;
    if (!good) why = "ncclCommInitAll failed";
    for (int i = 0; good && i < n; ++i) {
        good = hipSetDevice(devs[(size_t)i]) == hipSuccess && hipStreamCreateWithFlags(&cs.strm[(size_t)i], hipStreamNonBlocking) == hipSuccess &&
               hipMalloc((void **)&cs.dbuf[(size_t)i], 16) == hipSuccess;
        if (!good) why = "device buffer for the all-reduce";
    }
    if (!good) {
        for (int i = 0; i < n; ++i) { if (cs.dbuf[(size_t)i] || cs.strm[(size_t)i]) { (void)hipSetDevice(devs[(size_t)i]); if (cs.dbuf[(size_t)i]) (void)hipFree(cs.dbuf[(size_t)i]); if (cs.strm[(size_t)i]) (void)hipStreamDestroy(cs.strm[(size_t)i]); } if (cs.comms[(size_t)i]) R.CommDestroy(cs.comms[(size_t)i]); }
        return nullptr;
    }
    cs.init_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++g_comm_inits;
    return &(cache[devs] = std::move(cs));
}

// Lexicographic minimum of the devices' (hi, lo) words, left in every w[i]: ONE MIN all-reduce of hi, then one of lo among the
// ranks that hold the minimal hi (the others contribute all ones) -- the two-stage form of plinopt_amd/dist.py allreduce_best, so
// that no cost is too wide for the word.  `seconds` = wall time of the two all-reduces (communicator set-up excluded: it is
// cached).  false when RCCL is not there or a call fails (the caller keeps the host minimum and says so).
bool rccl_min_pair(const std::vector<int> &devs, std::vector<std::array<unsigned long long, 2>> &w, double &seconds, std::string &why)
{
    Rccl &R = rccl();
    if (!R.ok) { why = "librccl not loadable"; return false; }
    std::lock_guard<std::mutex> lk(g_comm_mu);
    CommSet *cs = comm_for(devs, why);
    if (!cs) return false;
    const int n = (int)devs.size();
    const auto t0 = std::chrono::steady_clock::now();
    auto stage = [&](int k) -> bool {                                              // all-reduce of word k of every device
        bool good = true;
        for (int i = 0; good && i < n; ++i) good = hipSetDevice(devs[(size_t)i]) == hipSuccess && hipMemcpyAsync(cs->dbuf[(size_t)i], &w[(size_t)i][(size_t)k], 8, hipMemcpyHostToDevice, cs->strm[(size_t)i]) == hipSuccess;
        if (!good) { why = "upload of the packed word"; return false; }
        R.GroupStart();
        for (int i = 0; i < n; ++i) good = R.AllReduce(cs->dbuf[(size_t)i], cs->dbuf[(size_t)i], 1, RCCL_UINT64, RCCL_MIN, cs->comms[(size_t)i], cs->strm[(size_t)i]) == 0 && good;
        good = (R.GroupEnd() == 0) && good;
        if (!good) { why = "ncclAllReduce failed"; return false; }
        for (int i = 0; good && i < n; ++i) good = hipSetDevice(devs[(size_t)i]) == hipSuccess && hipMemcpyAsync(&w[(size_t)i][(size_t)k], cs->dbuf[(size_t)i], 8, hipMemcpyDeviceToHost, cs->strm[(size_t)i]) == hipSuccess && hipStreamSynchronize(cs->strm[(size_t)i]) == hipSuccess;
        if (!good) why = "reading the reduced word";
        return good;
    };
    std::vector<unsigned long long> hi0((size_t)n);
    for (int i = 0; i < n; ++i) hi0[(size_t)i] = w[(size_t)i][0];
    bool good = stage(0);
    if (good) { for (int i = 0; i < n; ++i) if (hi0[(size_t)i] != w[(size_t)i][0]) w[(size_t)i][1] = ~0ull; good = stage(1); }
    seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return good;
}

// what a shard hands back: its best candidate as a (hi, lo) word pair ordered like the search's total order
struct MultiShard { int rc = PLO_OK; std::string msg; uint64_t s0 = 0, cnt = 0; plo_stats_t st{}; unsigned long long hi = ~0ull, lo = ~0ull; };

// body(shard, device ordinal) runs on its own host thread with its own device context, between plo_init and plo_shutdown
int multi_run(int ndev, const int *devices, uint64_t seed0, uint64_t nseeds, std::vector<MultiShard> &sh, const std::function<int(MultiShard &, int)> &body)
{
    sh.assign((size_t)ndev, MultiShard{});
    std::vector<std::thread> th;
    for (int r = 0; r < ndev; ++r) th.emplace_back([&, r]() {
        MultiShard &S = sh[(size_t)r];
        const uint64_t q = nseeds / (uint64_t)ndev, rem = nseeds % (uint64_t)ndev;                // the same blocks as the forked shards of the tools
        S.s0 = seed0 + (uint64_t)r * q + std::min<uint64_t>((uint64_t)r, rem); S.cnt = q + ((uint64_t)r < rem ? 1 : 0);
        if (S.cnt == 0) return;
        const int dev = devices ? devices[r] : r;
        DevCtx ctx; t_ctx = &ctx;                                                                 // this thread's device, stream and limits
        S.rc = plo_init(dev);
        if (S.rc == PLO_OK) S.rc = body(S, dev);
        if (S.rc != PLO_OK) S.msg = "device " + std::to_string(dev) + ": " + g_err;
        plo_shutdown();
        t_ctx = nullptr;
    });
    for (auto &t : th) t.join();
    // a refusal (PLO_E_CAPACITY / PLO_E_UNSUPPORTED) is reported as such when every failed shard says so: the caller may fall back
    for (auto &S : sh) if (S.rc != PLO_OK && S.rc != PLO_E_CAPACITY && S.rc != PLO_E_UNSUPPORTED) return fail(S.rc, S.msg);
    for (auto &S : sh) if (S.rc != PLO_OK) return fail(S.rc, S.msg);
    return PLO_OK;
}

// The minimum of the shards' words: on the host, and -- with two or more DISTINCT devices (PLO_MULTI_REDUCE=rccl: also for one;
// =host: never) -- by rccl_min_pair over the devices.  The all-reduce result is the one used; the host minimum is its check, and
// a difference is a diagnostic on stderr with the host minimum kept (never an error: the host value is right by construction).
// Returns the winning shard (-1: no shard has a candidate); agg gets the summed statistics, `reduce` and the all-reduce time.
int multi_min(const std::vector<MultiShard> &sh, int ndev, const int *devices, plo_stats_t &agg, int &winner)
{
    winner = -1;
    agg = plo_stats_t{};
    for (size_t r = 0; r < sh.size(); ++r) {
        const MultiShard &S = sh[r];
        if (S.cnt == 0) continue;
        agg.candidates += S.st.candidates; agg.launches += S.st.launches; agg.kernel_ms = std::max(agg.kernel_ms, S.st.kernel_ms);
        agg.grid = S.st.grid; agg.lds_bytes = S.st.lds_bytes; agg.waves_per_wg = S.st.waves_per_wg; agg.algo_bytes = S.st.algo_bytes;
        if (S.hi == ~0ull && S.lo == ~0ull) continue;
        if (winner < 0 || S.hi < sh[(size_t)winner].hi || (S.hi == sh[(size_t)winner].hi && S.lo < sh[(size_t)winner].lo)) winner = (int)r;
    }
    const char *mode = getenv("PLO_MULTI_REDUCE");
    const bool want = mode ? !strcmp(mode, "rccl") : ndev >= 2;
    if (!want || winner < 0) return PLO_OK;
    std::vector<int> devs; std::vector<std::array<unsigned long long, 2>> words;
    for (int r = 0; r < ndev; ++r) { devs.push_back(devices ? devices[r] : r); words.push_back({sh[(size_t)r].hi, sh[(size_t)r].lo}); }
    bool distinct = true; for (size_t i = 0; i < devs.size(); ++i) for (size_t j = 0; j < i; ++j) distinct = distinct && devs[i] != devs[j];
    if (!distinct) return PLO_OK;                                                  // (a communicator needs distinct devices: PLO_GPU_DEVICES=0,0 in the tests)
    std::string why; double secs = 0;
    if (rccl_min_pair(devs, words, secs, why)) {
        agg.reduce = 1; agg.reduce_seconds = secs;
        for (auto &w : words) if (w[0] != sh[(size_t)winner].hi || w[1] != sh[(size_t)winner].lo) {
            fprintf(stderr, "# libplinopt_hip: RCCL MIN all-reduce gave (%llx, %llx), the host minimum is (%llx, %llx): host minimum kept\n", w[0], w[1], sh[(size_t)winner].hi, sh[(size_t)winner].lo);
            agg.reduce = 0; break;
        }
    } else if (mode) return fail(PLO_E_HIP, std::string("PLO_MULTI_REDUCE=rccl: ") + why);
    return PLO_OK;
}
} // namespace

extern "C" {

uint64_t plo_multi_comm_inits(void) { return g_comm_inits; }

int plo_cse_search_multi(const plo_csr_t *A, uint32_t p, uint64_t seed0, uint64_t nseeds, int cost_mode, int ndev, const int *devices,
                         plo_best_t *out, plo_stats_t *stats)
{
    if (!A || !out) return fail(PLO_E_ARG, "null argument");
    if (ndev < 1 || ndev > 64) return fail(PLO_E_ARG, "device count outside [1,64]");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    DeviceGuard guard;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<MultiShard> sh; std::vector<plo_best_t> bests((size_t)ndev);
    auto key = [&](const plo_best_t &b) -> unsigned long long {                                   // cmpOpCount orders of plinopt_optimize.h:53-64
        switch (cost_mode) { case PLO_COST_ADD_THEN_MUL: return ((unsigned long long)b.adds << 32) | b.muls; case PLO_COST_SUM: return ((unsigned long long)b.adds + b.muls) << 32; default: return (((unsigned long long)b.adds + b.muls) << 32) | b.adds; } };
    int rc = multi_run(ndev, devices, seed0, nseeds, sh, [&](MultiShard &S, int) {
        plo_best_t &b = bests[(size_t)(&S - sh.data())];
        const int r_ = plo_cse_search(A, p, S.s0, S.cnt, cost_mode, &b, &S.st);
        if (r_ == PLO_OK && b.seed != ~0ull) { S.hi = key(b); S.lo = b.seed - seed0; }
        return r_;
    });
    if (rc != PLO_OK) return rc;
    plo_stats_t agg{}; int win = -1;
    rc = multi_min(sh, ndev, devices, agg, win);
    if (rc != PLO_OK) return rc;
    out->adds = out->muls = 0xFFFFFFFFu; out->seed = ~0ull;
    if (win >= 0) *out = bests[(size_t)win];
    agg.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = agg;
    return PLO_OK;
}

int plo_cse_chain_destroy(plo_chain_t *ch)
{
    if (!ch) return PLO_OK;
    for (int k = 0; k < 2; ++k) plo_cse_plan_destroy(ch->st[k]);
    delete ch;
    return PLO_OK;
}

int plo_cse_chain_create(const plo_csr_t *first, const plo_csr_t *second, uint32_t p, plo_chain_t **out)
{
    if (!first || !second || !out) return fail(PLO_E_ARG, "null argument");
    plo_chain *ch = new plo_chain();
    int rc = plo_cse_plan_create(first, p, &ch->st[0]);
    if (rc == PLO_OK) rc = plo_cse_plan_create(second, p, &ch->st[1]);
    if (rc == PLO_OK && (ch->st[0]->big || ch->st[1]->big)) rc = fail(PLO_E_CAPACITY, "chained search needs both matrices on the LDS-resident wave kernel");
    if (rc == PLO_OK) rc = chain_config(ch);
    if (rc != PLO_OK) { plo_cse_chain_destroy(ch); return rc; }
    *out = ch;
    return PLO_OK;
}

// Many pairs in one launch (plo::cse_chain_batch_kernel): candidate c = pair c / per_pair, seed seed0 + c.
int plo_cse_chain_batch(uint32_t npairs, const plo_csr_t *firsts, const plo_csr_t *seconds, uint32_t p, uint64_t seed0, uint32_t per_pair,
                        int cost_mode, uint32_t *adds, uint32_t *muls, plo_best_t *best, plo_stats_t *st)
{
    if (!firsts || !seconds || npairs == 0 || per_pair == 0) return fail(PLO_E_ARG, "bad argument");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    if (p < 3 || p >= 0x80000000u || !(p & 1u)) return fail(PLO_E_ARG, "modulus must be an odd prime below 2^31");
    for (uint32_t k = 0; k < npairs; ++k)
        for (const plo_csr_t *A : {&firsts[k], &seconds[k]})
            if (const char *bad = csr_defect(A, p)) return fail(PLO_E_ARG, "pair " + std::to_string(k) + ": " + bad);
    const uint64_t ncand = (uint64_t)npairs * per_pair;
    if (ncand > 0xFFFFFFFFull) return fail(PLO_E_ARG, "at most 2^32-1 candidates per call");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    const auto t0 = std::chrono::steady_clock::now();
    if (best) { best->adds = best->muls = 0xFFFFFFFFu; best->seed = ~0ull; }
    for (uint32_t cap_scale = 2; cap_scale <= 16; cap_scale *= 2) {
        // host part of all plans (OpenMP), one device buffer for all images
        std::vector<plo::WavePlan> plans(2ull * npairs);
        std::vector<std::vector<uint8_t>> imgs(2ull * npairs);
        std::vector<int> rcs(2ull * npairs, PLO_OK); std::vector<std::string> errs(2ull * npairs);
        {   // host threads share the plans out (no OpenMP runtime in this library: the callers bring their own)
            std::atomic<long long> next{0};
            auto work = [&]() {
                for (;;) {
                    const long long k = next.fetch_add(1);
                    if (k >= 2ll * npairs) break;
                    const plo_csr_t *A = (k & 1) ? &seconds[k >> 1] : &firsts[k >> 1];
                    plo_plan tmp; tmp.m = A->m; tmp.n = A->n; tmp.p = p; tmp.cap_scale = cap_scale;
                    if (!A->rowptr || (A->rowptr[A->m] && (!A->col || !A->val))) { rcs[k] = PLO_E_ARG; errs[k] = "null matrix arrays"; continue; }
                    tmp.rowptr.assign(A->rowptr, A->rowptr + A->m + 1);
                    tmp.col.assign(A->col, A->col + A->rowptr[A->m]); tmp.val.assign(A->val, A->val + A->rowptr[A->m]);
                    rcs[k] = build_plan(&tmp, &imgs[k]);
                    if (rcs[k] != PLO_OK) errs[k] = g_err; else plans[k] = tmp.P;
                }
            };
            const unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), std::min<unsigned>(64u, npairs)));
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
            work();
            for (auto &th : pool) th.join();
        }
        for (size_t k = 0; k < rcs.size(); ++k) if (rcs[k] != PLO_OK) return fail(rcs[k], "pair " + std::to_string(k >> 1) + ": " + errs[k]);
        uint32_t region = 0, rsmax = 0; size_t total = 0; std::vector<size_t> offs(imgs.size());
        for (size_t k = 0; k < imgs.size(); ++k) { region = std::max(region, plans[k].region_bytes); rsmax = std::max(rsmax, plans[k].rs_bytes); offs[k] = total; total += round_up((uint32_t)imgs[k].size(), 64); }
        uint32_t W = 0, lds = 0;
        for (uint32_t w : {4u, 2u, 1u}) { const uint32_t l = 64u + w * (2u * rsmax + region); if (l <= g_lds_max) { W = w; lds = l; break; } }
        if (!W) return fail(PLO_E_CAPACITY, "chained candidate state does not fit LDS");
        uint8_t *d_img = nullptr; plo::WavePlan *d_plans = nullptr; uint32_t *d_adds = nullptr, *d_muls = nullptr, *d_err = nullptr; unsigned long long *d_best = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<uint8_t> blob(total + 64, 0);
        for (size_t k = 0; k < imgs.size(); ++k) std::memcpy(blob.data() + offs[k], imgs[k].data(), imgs[k].size());
        auto cleanup = [&]() { if (d_img) (void)hipFree(d_img); if (d_plans) (void)hipFree(d_plans); if (d_adds) (void)hipFree(d_adds); if (d_muls) (void)hipFree(d_muls); if (d_err) (void)hipFree(d_err); if (d_best) (void)hipFree(d_best); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); e0 = e1 = nullptr; };
#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(PLO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
        BCHK(hipMalloc((void **)&d_img, blob.size()));
        BCHK(hipMemcpy(d_img, blob.data(), blob.size(), hipMemcpyHostToDevice));
        for (size_t k = 0; k < plans.size(); ++k) plans[k].tmpl = (const uint64_t *)(d_img + offs[k]);
        BCHK(hipMalloc((void **)&d_plans, plans.size() * sizeof(plo::WavePlan)));
        BCHK(hipMemcpy(d_plans, plans.data(), plans.size() * sizeof(plo::WavePlan), hipMemcpyHostToDevice));
        BCHK(hipMalloc((void **)&d_err, 4)); BCHK(hipMemsetAsync(d_err, 0, 4, g_stream));
        BCHK(hipMalloc((void **)&d_best, 8)); BCHK(hipMemsetAsync(d_best, 0xFF, 8, g_stream));
        if (adds) BCHK(hipMalloc((void **)&d_adds, ncand * 4));
        if (muls) BCHK(hipMalloc((void **)&d_muls, ncand * 4));
        BCHK(hipFuncSetAttribute((const void *)plo::cse_chain_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        BCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)plo::cse_chain_batch_kernel, (int)(W * 64), lds));
        const uint64_t need = (ncand + W - 1) / W;
        const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g_cus * std::max(nb, 1), need));
        plo::WaveJob J{}; J.seed0 = seed0; J.seeds = nullptr; J.ncand = ncand; J.adds = d_adds; J.muls = d_muls; J.best = best ? d_best : nullptr; J.cost_mode = (uint32_t)cost_mode; J.err = d_err;
        BCHK(hipEventCreate(&e0)); BCHK(hipEventCreate(&e1));
        BCHK(hipEventRecord(e0, g_stream));
        hipLaunchKernelGGL(plo::cse_chain_batch_kernel, dim3((uint32_t)grid), dim3(W * 64), lds, g_stream, (const plo::WavePlan *)d_plans, per_pair, region, rsmax, J);
        BCHK(hipGetLastError());
        BCHK(hipEventRecord(e1, g_stream)); BCHK(hipEventSynchronize(e1));
        float ms = 0; BCHK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); e0 = e1 = nullptr;
        uint32_t err = 0; unsigned long long w = 0;
        BCHK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = lds; st->waves_per_wg = W; st->candidates = ncand;
        if (err == plo::ERR_TABLE) { cleanup(); continue; }                         // a pair table filled up: all plans again with twice the slots
        if (err) { cleanup(); return device_error((int)err); }
        if (adds) BCHK(hipMemcpy(adds, d_adds, ncand * 4, hipMemcpyDeviceToHost));
        if (muls) BCHK(hipMemcpy(muls, d_muls, ncand * 4, hipMemcpyDeviceToHost));
        if (best) {
            BCHK(hipMemcpy(&w, d_best, 8, hipMemcpyDeviceToHost));
            const uint64_t key = w >> 32, off = w & 0xFFFFFFFFull;
            uint32_t a = 0, mu = 0;
            if (cost_mode == PLO_COST_SUM_THEN_ADD) { a = (uint32_t)(key & 0xFFFFu); mu = (uint32_t)(key >> 16) - a; }
            else if (cost_mode == PLO_COST_ADD_THEN_MUL) { a = (uint32_t)(key >> 16); mu = (uint32_t)(key & 0xFFFFu); }
            else { a = (uint32_t)key; mu = 0; }                                     // sum only: reported in .adds
            best->adds = a; best->muls = mu; best->seed = seed0 + off;
        }
#undef BCHK
        cleanup();
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return PLO_OK;
    }
    return device_error(plo::ERR_TABLE);
}

// ---------------------------------------------------------------------------------------------------------------
// The kernel method with everything on the device (plo_kmethod.hip): decomposition, both images and both Optimizer
// calls of every restart in one wave.
namespace {
uint32_t rank_mod_p(const plo_csr_t *M, uint32_t p)
{
    const uint32_t m = M->m, n = M->n;
    std::vector<std::vector<uint64_t>> A(m, std::vector<uint64_t>(n, 0));
    for (uint32_t i = 0; i < m; ++i) for (uint32_t k = M->rowptr[i]; k < M->rowptr[i + 1]; ++k) A[i][M->col[k]] = M->val[k];
    uint32_t r = 0;
    for (uint32_t c = 0; c < n && r < m; ++c) {
        uint32_t piv = r; while (piv < m && A[piv][c] == 0) ++piv;
        if (piv == m) continue;
        std::swap(A[piv], A[r]);
        const uint64_t iv = inv_mod((uint32_t)A[r][c], p);
        for (uint32_t j = c; j < n; ++j) A[r][j] = A[r][j] * iv % p;
        for (uint32_t i = r + 1; i < m; ++i) if (A[i][c]) { const uint64_t x = A[i][c]; for (uint32_t j = c; j < n; ++j) A[i][j] = (A[i][j] + (p - x) * A[r][j]) % p; }
        ++r;
    }
    return r;
}
// layout of an image that is built on the device (no template): the fields of build_plan() from bounds instead of a matrix
int layout_plan(plo::WavePlan &P, uint32_t m, uint32_t n, uint32_t nnz, uint32_t p, uint32_t maxlen, uint32_t naive, uint32_t cap)
{
    P = plo::WavePlan{};
    const uint32_t mw = 1;
    const uint64_t NC = (uint64_t)n + naive / 2 + 2;
    const uint32_t rb = ceil_log2(p), bb = ceil_log2((uint32_t)NC);
    if (NC >= 0xFFFFull || 2u * bb + rb > 51u) return fail(PLO_E_CAPACITY, "pair key (col,col,ratio) of the dependent part does not fit 51 bits");
    if (2ull * nnz >= 65535ull || cap > 65536u) return fail(PLO_E_CAPACITY, "dependent part too large for the wave kernel");
    P.m = m; P.n = n; P.nnz = nnz; P.p = p; P.NC = (uint32_t)NC; P.cap = cap; P.hbits = ceil_log2(cap);
    P.lpr_log2 = std::max(2u, ceil_log2(std::max(maxlen, 1u))); P.mw = mw; P.unit = 0u;
    P.multcap = (uint32_t)(naive / 2 + 8); P.maxlen = maxlen; P.rb = rb; P.bb = bb; P.mu = (~0ull) / p;
    uint32_t off = 0;
    P.off_tab = off;   off += cap * 8u;
    P.off_val = off;   off += nnz * 4u;
    P.off_inv = off;   off += nnz * 4u;
    P.off_col = off;   off += nnz * 2u;
    P.off_len = off;   off += m * 2u;
    off = round_up(off, 8);
    P.off_cmask = off; P.off_umask = off + mw * 8u;
    P.tmpl_bytes = off + n * 2u * mw * 8u;
    off += (uint32_t)NC * 2u * mw * 8u;
    P.off_aff = off;   off += (2u * mw + 1u) * 8u;
    P.off_ties = off;  off += cap * 2u;
    off = round_up(off, 8);
    P.off_mult = off;  off += P.multcap * 8u;
    P.region_bytes = round_up(off, 16);
    P.rs_bytes = round_up((m + 1) * 2u, 16);
    return PLO_OK;
}
} // namespace

int plo_kernel_search(const plo_csr_t *M, uint32_t p, uint64_t seed0, uint64_t nrestarts, uint32_t per_block, int cost_mode,
                      uint32_t *adds, uint32_t *muls, uint32_t *info, plo_best_t *best, plo_stats_t *st)
{
    if (!M || nrestarts == 0 || per_block == 0) return fail(PLO_E_ARG, "bad argument");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init not called");
    if (p < 3 || p >= 0x80000000u || !(p & 1u)) return fail(PLO_E_ARG, "modulus must be an odd prime below 2^31");
    if (const char *bad = csr_defect(M, p)) return fail(PLO_E_ARG, bad);
    if (M->m == 0 || M->m > 128 || M->n == 0 || M->n > 64) return fail(PLO_E_UNSUPPORTED, "device decomposition needs at most 128 rows and 64 columns");
    if (nrestarts > 0xFFFFFFFFull) return fail(PLO_E_ARG, "at most 2^32-1 restarts per call");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    const auto t0 = std::chrono::steady_clock::now();
    if (best) { best->adds = best->muls = 0xFFFFFFFFu; best->seed = ~0ull; }
    const uint32_t m = M->m, n = M->n, R = rank_mod_p(M, p), ndeps = m - R;
    if (ndeps == 0) return fail(PLO_E_UNSUPPORTED, "zero dimensional kernel");
    if (ndeps > 64) return fail(PLO_E_UNSUPPORTED, "more than 64 dependent rows: Dep's ProgramGen keeps one row per lane");
    if (R == 0) return fail(PLO_E_UNSUPPORTED, "zero matrix");

    uint8_t *d_img = nullptr; uint64_t *d_rsD = nullptr; uint32_t *d_adds = nullptr, *d_muls = nullptr, *d_info = nullptr, *d_err = nullptr, *d_sz = nullptr; unsigned long long *d_best = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_img, (void *)d_rsD, (void *)d_adds, (void *)d_muls, (void *)d_info, (void *)d_err, (void *)d_sz, (void *)d_best}) if (q) (void)hipFree(q);
        d_img = nullptr; d_rsD = nullptr; d_adds = d_muls = d_info = d_err = d_sz = nullptr; d_best = nullptr;
        if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); e0 = e1 = nullptr;
    };
#define KCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(PLO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    uint32_t pairs_max = 0, ent_max = 0;
    // Dep's table starts at 0.75 x the sampled PAIR count (an upper bound of its distinct triples, usually far above it): a
    // smaller image means more resident waves; a full table is reported by the device and the launch repeated with twice the slots
    for (uint32_t cap_scale = 1; cap_scale <= 32; cap_scale *= 2) {
        plo::KPlan K{};
        // plan and image of M (the layout of Free)
        plo_plan tmp; tmp.m = m; tmp.n = n; tmp.p = p; tmp.cap_scale = std::max(2u, cap_scale);
        tmp.rowptr.assign(M->rowptr, M->rowptr + m + 1); tmp.col.assign(M->col, M->col + M->rowptr[m]); tmp.val.assign(M->val, M->val + M->rowptr[m]);
        std::vector<uint8_t> img;
        int rc = build_plan(&tmp, &img);
        if (rc != PLO_OK) { cleanup(); return rc; }
        K.PM = tmp.P; K.m = m; K.n = n; K.rank = R; K.ndeps = ndeps; K.per_block = per_block;
        K.mers = 0; for (uint32_t kk = 2; kk < 31; ++kk) if (p == (1u << kk) - 1u) K.mers = kk;
        uint32_t off = 0;
        const uint32_t depc_bytes = round_up(ndeps * R * 4u, 16);
        const bool depc_inside = !getenv("PLO_KMETHOD_DEPC_SCRATCH");             // (A/B knob: the round-3 placement in the scratch)
        if (!depc_inside) { K.off_depc = (int32_t)off; off += depc_bytes; }
        K.off_vrow = off;  off += 64u * 4u;
        K.off_ord = off;   off += 128u * 2u;
        K.off_piv = off;   off += 128u * 2u;
        K.off_basis = off; off += 64u * 2u;
        K.off_deps = off;  off += 64u * 2u;
        K.off_rsd = off;   off += 66u * 2u;
        K.scratch_bytes = round_up(off, 16);
        K.off_vc = round_up(K.PM.tmpl_bytes, 16);                                 // V (m x (n+1)) and C (m x (rank+1)) of the elimination, behind M's template
        const uint32_t vc_end = K.off_vc + m * (n + 1u + R + 1u) * 4u;
        KCHK(hipMalloc((void **)&d_img, img.size() + 64)); KCHK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
        K.PM.tmpl = (const uint64_t *)d_img;
        KCHK(hipMalloc((void **)&d_err, 4)); KCHK(hipMemsetAsync(d_err, 0, 4, g_stream));
        KCHK(hipEventCreate(&e0)); KCHK(hipEventCreate(&e1));
        if (cap_scale == 1) {
            // sizing launch: Dep's pair count over a sample of the decompositions
            K.region = round_up(std::max(K.PM.region_bytes, vc_end), 16);
            if (depc_inside) { const uint32_t at = round_up(std::max(K.PM.region_bytes, vc_end), 16); K.region = at + depc_bytes; K.off_depc = -(int32_t)depc_bytes; }
            uint32_t W = 0, lds = 0;
            for (uint32_t w : {4u, 2u, 1u}) { const uint32_t l = K.PM.rs_bytes + w * (K.region + K.scratch_bytes); if (l <= g_lds_max) { W = w; lds = l; break; } }
            if (!W) { cleanup(); return fail(PLO_E_CAPACITY, "kernel-method state does not fit LDS"); }
            KCHK(hipMalloc((void **)&d_sz, 16)); KCHK(hipMemsetAsync(d_sz, 0, 16, g_stream));
            KCHK(hipFuncSetAttribute((const void *)plo::kmethod_size_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            plo::WaveJob J{}; J.seed0 = seed0; J.ncand = std::min<uint64_t>(nrestarts, 4096); J.err = d_err;
            const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g_cus * 2u, (J.ncand + W - 1) / W));
            KCHK(hipEventRecord(e0, g_stream));
            hipLaunchKernelGGL(plo::kmethod_size_kernel, dim3((uint32_t)grid), dim3(W * 64), lds, g_stream, K, J, d_sz);
            KCHK(hipGetLastError());
            KCHK(hipEventRecord(e1, g_stream)); KCHK(hipEventSynchronize(e1));
            float ms = 0; KCHK(hipEventElapsedTime(&ms, e0, e1)); st->kernel_ms += ms; st->launches += 1;
            uint32_t sz[4] = {0, 0, 0, 0};
            KCHK(hipMemcpy(sz, d_sz, 16, hipMemcpyDeviceToHost));
            if (sz[2]) { cleanup(); return device_error(plo::ERR_KDEC); }
            pairs_max = sz[0]; ent_max = sz[1];
            if (const char *e = getenv("PLO_KMETHOD_PAIRS_DIV")) pairs_max /= (uint32_t)std::max(1l, strtol(e, nullptr, 10));   // test knob: undersized first table, exercises the repeat-with-more-slots path
        }
        // layout of Dep's image: rows <= m - rank, entries per row <= rank (hard bounds); the number of entries and the pair count from the
        // sample of the sizing launch (+60 %; measured on 4x4x4_49_156_L: +25 % and +40 % are exceeded by a few of 10^6 restarts and the whole launch is repeated, +60 % and +80 % never in 3 x 10^6: the rows of a restart are packed, a restart with more entries is reported -- ERR_TABLE -- and
        // the launch repeated with the hard bound rows x rank, as for a full table).  Round 2 sized the arrays for rows x rank: 17.3 KB per
        // wave on 4x4x4_49_156_L, 7 waves per CU.
        uint32_t capD = 64;
        while (capD < (cap_scale * pairs_max * 3u) / 4u + 16u) capD <<= 1;
        uint32_t entD = (cap_scale == 1 && !getenv("PLO_KMETHOD_HARD_BOUNDS")) ? std::min<uint32_t>(ndeps * R, ent_max + (uint32_t)((uint64_t)ent_max * (getenv("PLO_KMETHOD_ENT_MARGIN") ? (uint32_t)atoi(getenv("PLO_KMETHOD_ENT_MARGIN")) : 60u) / 100u) + 16u) : ndeps * R;
        if (const char *e = getenv("PLO_KMETHOD_ENT_DIV")) { if (cap_scale == 1) entD = std::max<uint32_t>(R, entD / (uint32_t)std::max(1l, strtol(e, nullptr, 10))); }   // test knob: undersized entry arrays, exercises the repeat-with-hard-bounds path
        rc = layout_plan(K.PD, ndeps, m, entD, p, R, entD, capD);
        if (rc != PLO_OK) { cleanup(); return rc; }
        K.rsD = nullptr;                                                          // Dep's row starts are computed per restart on the device
        K.region = round_up(std::max(std::max(K.PM.region_bytes, K.PD.region_bytes), vc_end), 16);
        if (depc_inside) {
            // the combinations live from the end of the decomposition to the end of Dep's image build: behind Free's region (Optimizer on Free
            // runs meanwhile) and behind the elimination arrays (they are copied out of C), over Dep's tie list / multiplier list (idle until
            // Dep's Optimizer call)
            const uint32_t at = round_up(std::max(std::max(K.PM.region_bytes, vc_end), K.PD.off_ties), 16);
            K.region = std::max(K.region, at + depc_bytes);
            K.off_depc = (int32_t)at - (int32_t)K.region;
        }
        uint32_t W = 0, lds = 0, bestw = 0;
        for (uint32_t w : {4u, 2u, 1u}) {
            const uint32_t l = K.PM.rs_bytes + K.PD.rs_bytes + w * (K.region + K.scratch_bytes);
            if (l > g_lds_max) continue;
            const uint32_t waves = std::min<uint32_t>(32u, (uint32_t)(g_lds_max / l) * w);
            if (waves > bestw) { bestw = waves; W = w; lds = l; }
        }
        if (!W) { cleanup(); return fail(PLO_E_CAPACITY, "kernel-method state does not fit LDS"); }
        if (getenv("PLO_KM_DEBUG")) fprintf(stderr, "# kernel method layout: region of M %u B (table %u slots), of Dep %u B (table %u slots, sampled pairs %u, entries %u of at most %u), elimination arrays end at %u B, scratch %u B; %u waves per workgroup, %u B of LDS, %u waves per CU\n",
                                            K.PM.region_bytes, K.PM.cap, K.PD.region_bytes, K.PD.cap, pairs_max, K.PD.nnz, ndeps * R, vc_end, K.scratch_bytes, W, lds, bestw);
        KCHK(hipMalloc((void **)&d_best, 8)); KCHK(hipMemsetAsync(d_best, 0xFF, 8, g_stream));
        if (adds) KCHK(hipMalloc((void **)&d_adds, nrestarts * 4));
        if (muls) KCHK(hipMalloc((void **)&d_muls, nrestarts * 4));
        if (info) KCHK(hipMalloc((void **)&d_info, nrestarts * 12));
        const void *fn = K.PM.unit ? (const void *)plo::kmethod_kernel<true> : (const void *)plo::kmethod_kernel<false>;
        KCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        KCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (int)(W * 64), lds));
        const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g_cus * std::max(nb, 1), (nrestarts + W - 1) / W));
        plo::WaveJob J{}; J.seed0 = seed0; J.ncand = nrestarts; J.adds = d_adds; J.muls = d_muls; J.best = best ? d_best : nullptr; J.cost_mode = (uint32_t)cost_mode; J.err = d_err;
        plo::KInfo I{d_info};
        KCHK(hipEventRecord(e0, g_stream));
        if (K.PM.unit) hipLaunchKernelGGL(plo::kmethod_kernel<true>, dim3((uint32_t)grid), dim3(W * 64), lds, g_stream, K, J, I);
        else hipLaunchKernelGGL(plo::kmethod_kernel<false>, dim3((uint32_t)grid), dim3(W * 64), lds, g_stream, K, J, I);
        KCHK(hipGetLastError());
        KCHK(hipEventRecord(e1, g_stream)); KCHK(hipEventSynchronize(e1));
        float ms = 0; KCHK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t err = 0; KCHK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = lds; st->waves_per_wg = W; st->candidates = nrestarts;
#ifdef PLO_KM_PROFILE
        if (getenv("PLO_KM_STATS")) {
            unsigned long long g[8] = {0};
            if (hipMemcpyFromSymbol(g, HIP_SYMBOL(plo::g_kprof), sizeof g) == hipSuccess && g[5])
                fprintf(stderr, "# kernel method, cumulative over %llu restarts: cycles per restart: copy + decomposition %.0f, Free image %.0f, Optimizer on Free %.0f, Dep image %.0f, Optimizer on Dep %.0f; inside the decomposition: elimination loops %.0f, pivot + inverse + stores %.0f\n",
                        g[5], (double)g[0] / g[5], (double)g[1] / g[5], (double)g[2] / g[5], (double)g[3] / g[5], (double)g[4] / g[5], (double)g[6] / g[5], (double)g[7] / g[5]);
        }
#endif
        if (err == plo::ERR_TABLE) { cleanup(); continue; }                      // a pair table filled up: again with twice the slots
        if (err) { cleanup(); return device_error((int)err); }
        if (adds) KCHK(hipMemcpy(adds, d_adds, nrestarts * 4, hipMemcpyDeviceToHost));
        if (muls) KCHK(hipMemcpy(muls, d_muls, nrestarts * 4, hipMemcpyDeviceToHost));
        if (info) KCHK(hipMemcpy(info, d_info, nrestarts * 12, hipMemcpyDeviceToHost));
        if (best) {
            unsigned long long w = 0;
            KCHK(hipMemcpy(&w, d_best, 8, hipMemcpyDeviceToHost));
            const uint64_t key = w >> 32, o = w & 0xFFFFFFFFull;
            uint32_t a = 0, mu = 0;
            if (cost_mode == PLO_COST_SUM_THEN_ADD) { a = (uint32_t)(key & 0xFFFFu); mu = (uint32_t)(key >> 16) - a; }
            else if (cost_mode == PLO_COST_ADD_THEN_MUL) { a = (uint32_t)(key >> 16); mu = (uint32_t)(key & 0xFFFFu); }
            else { a = (uint32_t)key; mu = 0; }
            best->adds = a; best->muls = mu; best->seed = seed0 + o;
        }
        cleanup();
        st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return PLO_OK;
    }
#undef KCHK
    return device_error(plo::ERR_TABLE);
}

int plo_cse_chain_cost_many(plo_chain_t *ch, const uint64_t *seeds, uint64_t seed0, uint64_t n,
                            uint32_t *adds, uint32_t *muls, plo_stats_t *st)
{
    if (!ch || !adds || !muls) return fail(PLO_E_ARG, "null argument");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    if (n == 0) return PLO_OK;
    uint32_t *d_adds = nullptr, *d_muls = nullptr; uint64_t *d_seeds = nullptr;
    HIPCHK(hipMalloc((void **)&d_adds, n * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&d_muls, n * sizeof(uint32_t)));
    if (seeds) { HIPCHK(hipMalloc((void **)&d_seeds, n * sizeof(uint64_t))); HIPCHK(hipMemcpy(d_seeds, seeds, n * sizeof(uint64_t), hipMemcpyHostToDevice)); }
    plo::WaveJob J{}; J.seed0 = seed0; J.seeds = d_seeds; J.ncand = n; J.adds = d_adds; J.muls = d_muls; J.best = nullptr; J.cost_mode = 0;
    int rc = run_chain(ch, J, st);
    if (rc == PLO_OK) {
        hipError_t e1 = hipMemcpy(adds, d_adds, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(muls, d_muls, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(PLO_E_HIP, "copy back failed");
    }
    (void)hipFree(d_adds); (void)hipFree(d_muls); if (d_seeds) (void)hipFree(d_seeds);
    st->candidates = n;
    return rc;
}

int plo_cse_chain_search(plo_chain_t *ch, uint64_t seed0, uint64_t nseeds, int cost_mode, plo_best_t *out, plo_stats_t *st)
{
    if (!ch || !out) return fail(PLO_E_ARG, "null argument");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    out->adds = out->muls = 0xFFFFFFFFu; out->seed = ~0ull;
    uint64_t bkey = ~0ull, bseed = ~0ull;
    plo_plan *p0 = ch->st[0];
    for (uint64_t done = 0; done < nseeds;) {
        const uint64_t cnt = std::min<uint64_t>(0xFFFFFFFFull, nseeds - done);
        HIPCHK(hipMemsetAsync(p0->d_best, 0xFF, sizeof(unsigned long long), g_stream));
        plo::WaveJob J{}; J.seed0 = seed0 + done; J.ncand = cnt; J.best = p0->d_best; J.cost_mode = (uint32_t)cost_mode;
        int rc = run_chain(ch, J, st);
        if (rc != PLO_OK) return rc;
        unsigned long long w = 0;
        HIPCHK(hipMemcpy(&w, p0->d_best, sizeof w, hipMemcpyDeviceToHost));
        const uint64_t key = w >> 32, sd = seed0 + done + (w & 0xFFFFFFFFull);
        if (key < bkey || (key == bkey && sd < bseed)) { bkey = key; bseed = sd; }
        done += cnt;
    }
    if (nseeds) {
        uint32_t a = 0, mu = 0; plo_stats_t s2{};
        int rc = plo_cse_chain_cost_many(ch, &bseed, 0, 1, &a, &mu, &s2);
        if (rc != PLO_OK) return rc;
        out->adds = a; out->muls = mu; out->seed = bseed;
    }
    st->candidates = nseeds;
    return PLO_OK;
}

int plo_cob_search(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock,
                   const uint32_t *coeffs, uint32_t ncoeffs, uint32_t p, int32_t w0, int32_t w1,
                   plo_cob_best_t *out, plo_stats_t *st)
{
    return plo_cob_search_range(n, m, TM, Cand, row, offsetblock, coeffs, ncoeffs, p, w0, w1, 0, (uint64_t)ncoeffs * ncoeffs * ncoeffs, out, st);
}

namespace {
// Host side of one enumeration: the right nullspace of the rows already chosen restricted to the block, the block of TM, the
// initial best word.  dependent: the chosen rows are dependent (rank(Cand with row := w) can never exceed `row`, :174).
struct CobPrep { bool dependent = false; uint32_t qn = 0, fb = 0; std::vector<uint32_t> nb, tmb; unsigned long long init = 0; };
int cob_check(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock, const uint32_t *coeffs, uint32_t ncoeffs, uint32_t p)
{
    if (!TM || !Cand || !coeffs) return fail(PLO_E_ARG, "null argument");
    if (p < 3 || p >= 0x80000000u || !(p & 1u)) return fail(PLO_E_ARG, "modulus must be an odd prime below 2^31");
    if (n == 0 || row >= n || offsetblock >= n || ncoeffs == 0) return fail(PLO_E_ARG, "bad dimensions");
    if (ncoeffs > 255) return fail(PLO_E_CAPACITY, "more than 255 coefficients: the candidate index does not fit 32 bits");
    if ((uint64_t)(m + 1) * (n + 1) >= 0xFFFFFFFFull) return fail(PLO_E_CAPACITY, "score does not fit 32 bits");
    return PLO_OK;
}
void cob_prepare(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock, uint32_t p, int32_t w0, int32_t w1, CobPrep &R)
{
    // right nullspace of the rows already chosen (rows 0..row-1 of Cand), by reduced row echelon form mod p
    std::vector<std::vector<uint32_t>> A(row, std::vector<uint32_t>(n));
    for (uint32_t i = 0; i < row; ++i) for (uint32_t j = 0; j < n; ++j) A[i][j] = Cand[(size_t)i * n + j] % p;
    std::vector<uint32_t> piv; uint32_t r0 = 0;
    for (uint32_t c = 0; c < n && r0 < row; ++c) {
        uint32_t q = r0; while (q < row && A[q][c] == 0) ++q;
        if (q == row) continue;
        std::swap(A[q], A[r0]);
        const uint32_t iv = inv_mod(A[r0][c], p);
        for (uint32_t j = c; j < n; ++j) A[r0][j] = (uint32_t)((uint64_t)A[r0][j] * iv % p);
        for (uint32_t i = 0; i < row; ++i) if (i != r0 && A[i][c]) { const uint32_t l = A[i][c]; for (uint32_t j = c; j < n; ++j) A[i][j] = (uint32_t)(((uint64_t)A[i][j] + (uint64_t)(p - l) * A[r0][j]) % p); }
        piv.push_back(c); ++r0;
    }
    R.dependent = piv.size() != row;
    if (R.dependent) return;
    std::vector<char> isp(n, 0); for (uint32_t c : piv) isp[c] = 1;
    const uint32_t fb = std::min<uint32_t>(4, n - offsetblock);
    std::vector<std::vector<uint32_t>> cols;                  // only the block positions of each basis vector
    for (uint32_t fc = 0; fc < n; ++fc) {
        if (isp[fc]) continue;
        std::vector<uint32_t> x(n, 0); x[fc] = 1;
        for (size_t k = 0; k < piv.size(); ++k) x[piv[k]] = A[k][fc] ? p - A[k][fc] : 0;
        bool any = false; for (uint32_t t = 0; t < fb; ++t) any |= x[offsetblock + t] != 0;
        if (any) cols.push_back({x[offsetblock], fb > 1 ? x[offsetblock + 1] : 0, fb > 2 ? x[offsetblock + 2] : 0, fb > 3 ? x[offsetblock + 3] : 0});
    }
    R.fb = fb; R.qn = (uint32_t)cols.size(); R.nb.assign(4 * (size_t)std::max<uint32_t>(R.qn, 1), 0);
    for (uint32_t c = 0; c < R.qn; ++c) for (uint32_t t = 0; t < 4; ++t) R.nb[(size_t)t * R.qn + c] = cols[c][t];
    R.tmb.assign(4 * (size_t)m, 0);
    for (uint32_t t = 0; t < fb; ++t) for (uint32_t j = 0; j < m; ++j) R.tmb[(size_t)t * m + j] = TM[(size_t)(offsetblock + t) * m + j] % p;
    const int64_t thr = w0 < 0 ? -1 : (int64_t)w0 * (n + 1) + std::max(w1, 0);
    R.init = ((unsigned long long)(thr + 1) << 32) | 0xFFFFFFFFull;
}
void cob_decode(unsigned long long w, unsigned long long init, uint32_t n, int32_t w0, int32_t w1, plo_cob_best_t *out)
{
    out->found = w != init ? 1u : 0u;
    if (out->found) { const uint32_t sc = (uint32_t)(w >> 32) - 1u; out->zeros_v = (int32_t)(sc / (n + 1)); out->zeros_w = (int32_t)(sc % (n + 1)); out->index = (uint32_t)~(uint32_t)w; }
    else { out->zeros_v = w0; out->zeros_w = w1; out->index = 0; }
}
} // namespace

// Up to PLO_COB_BATCH enumerations of the same shape (n, m, row, block) in ONE launch: one upload, one kernel (blockIdx.y = the
// enumeration), one download.  bin/sparsifier over the rationals enumerates modulo two primes: one launch instead of two.
int plo_cob_search_batch(uint32_t nprob, uint32_t n, uint32_t m, uint32_t row, uint32_t offsetblock, const plo_cob_problem_t *prob, plo_cob_best_t *out, plo_stats_t *st)
{
    if (!prob || !out || nprob == 0 || nprob > PLO_COB_BATCH) return fail(PLO_E_ARG, "1 to 4 enumerations per launch");
    for (uint32_t k = 0; k < nprob; ++k) { const int rc = cob_check(n, m, prob[k].TM, prob[k].Cand, row, offsetblock, prob[k].coeffs, prob[k].ncoeffs, prob[k].p); if (rc != PLO_OK) return rc; }
    if (g_device < 0) { int rc = plo_init(0); if (rc != PLO_OK) return rc; }
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    CobPrep R[PLO_COB_BATCH]; uint32_t live[PLO_COB_BATCH], nlive = 0;
    for (uint32_t k = 0; k < nprob; ++k) {
        cob_prepare(n, m, prob[k].TM, prob[k].Cand, row, offsetblock, prob[k].p, prob[k].w0, prob[k].w1, R[k]);
        const uint64_t C = prob[k].ncoeffs;
        st->candidates += C * C * C * C;
        if (R[k].dependent) { out[k].found = 0; out[k].zeros_v = prob[k].w0; out[k].zeros_w = prob[k].w1; out[k].index = 0; }
        else live[nlive++] = k;
    }
    if (nlive == 0) { st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); return PLO_OK; }
    // one device buffer kept by the context, one upload: [best words (2 each) | per enumeration: TM block | nullspace block | coefficients]
    DevCtx &cx = cur_ctx();
    size_t off[PLO_COB_BATCH][3], words = 2 * (size_t)nlive; bool tab = !getenv("PLO_COB_GENERIC"); size_t lds = 0;
    plo::CobBatch B{}; uint64_t maxgroups = 0, maxtotal = 0;
    for (uint32_t q = 0; q < nlive; ++q) {
        const uint32_t k = live[q], C = prob[k].ncoeffs;
        off[q][0] = words; words += R[k].tmb.size(); off[q][1] = words; words += R[k].nb.size(); off[q][2] = words; words += C;
        B.ms[q] = m | 1u; B.qs[q] = std::max<uint32_t>(R[k].qn, 1u) | 1u;
        tab = tab && 4ull * C * ((size_t)B.ms[q] + B.qs[q]) * 4 <= 72u * 1024u;
        maxgroups = std::max<uint64_t>(maxgroups, (uint64_t)C * C * C); maxtotal = std::max<uint64_t>(maxtotal, (uint64_t)C * C * C * C);
    }
    for (uint32_t q = 0; q < nlive; ++q) { const uint32_t k = live[q], C = prob[k].ncoeffs; lds = std::max<size_t>(lds, tab ? (size_t)(4ull * C * ((size_t)B.ms[q] + B.qs[q]) * 4) : (4 * (size_t)m + 4 * (size_t)R[k].qn + C) * 4); }
    if (lds > g_lds_max) return fail(PLO_E_CAPACITY, "block of TM does not fit LDS");
    if (cx.cob_words < words) {
        if (cx.cob_buf) (void)hipFree(cx.cob_buf);
        cx.cob_buf = nullptr; cx.cob_words = 0;
        HIPCHK(hipMalloc((void **)&cx.cob_buf, (words + 1024) * 4)); cx.cob_words = words + 1024;
    }
    if (!cx.cob_e0) { HIPCHK(hipEventCreate(&cx.cob_e0)); HIPCHK(hipEventCreate(&cx.cob_e1)); }
    std::vector<uint32_t> up(words);
    for (uint32_t q = 0; q < nlive; ++q) {
        const uint32_t k = live[q], C = prob[k].ncoeffs, p = prob[k].p;
        memcpy(up.data() + 2 * q, &R[k].init, 8);
        memcpy(up.data() + off[q][0], R[k].tmb.data(), R[k].tmb.size() * 4); memcpy(up.data() + off[q][1], R[k].nb.data(), R[k].nb.size() * 4); memcpy(up.data() + off[q][2], prob[k].coeffs, (size_t)C * 4);
        plo::CobJob &J = B.J[q];
        J.n = n; J.m = m; J.qn = R[k].qn; J.fb = R[k].fb; J.C = C; J.p = p; J.mu = (~0ull) / p; J.first = 0; J.total = (uint64_t)C * C * C * C;
        J.tm = cx.cob_buf + off[q][0]; J.nb = cx.cob_buf + off[q][1]; J.coeffs = cx.cob_buf + off[q][2]; J.best = (unsigned long long *)cx.cob_buf + q;
    }
    for (uint32_t q = nlive; q < PLO_COB_BATCH; ++q) B.J[q] = B.J[0];      // (never launched: blockIdx.y < nlive)
    B.tab = tab ? 1u : 0u;
    HIPCHK(hipMemcpyAsync(cx.cob_buf, up.data(), words * 4, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipFuncSetAttribute((const void *)plo::cob_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t grid = tab ? std::max<uint64_t>(1, std::min<uint64_t>((maxgroups + 3) / 4, (uint64_t)g_cus * 2)) : std::max<uint64_t>(1, std::min<uint64_t>((maxtotal + 255) / 256, (uint64_t)g_cus * 8));
    HIPCHK(hipEventRecord(cx.cob_e0, g_stream));
    hipLaunchKernelGGL(plo::cob_batch_kernel, dim3((uint32_t)grid, nlive), dim3(256), lds, g_stream, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(cx.cob_e1, g_stream)); HIPCHK(hipEventSynchronize(cx.cob_e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, cx.cob_e0, cx.cob_e1));
    unsigned long long w[PLO_COB_BATCH] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(w, cx.cob_buf, 8 * (size_t)nlive, hipMemcpyDeviceToHost));
    for (uint32_t q = 0; q < nlive; ++q) { const uint32_t k = live[q]; cob_decode(w[q], R[k].init, n, prob[k].w0, prob[k].w1, &out[k]); }
    st->kernel_ms = ms; st->launches = 1; st->grid = (uint32_t)grid; st->lds_bytes = (uint32_t)lds; st->waves_per_wg = 4;
    st->algo_bytes = (16ull * m + 8) * nlive;
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PLO_OK;
}

int plo_cob_search_range(uint32_t n, uint32_t m, const uint32_t *TM, const uint32_t *Cand, uint32_t row, uint32_t offsetblock,
                         const uint32_t *coeffs, uint32_t ncoeffs, uint32_t p, int32_t w0, int32_t w1,
                         uint64_t first_group, uint64_t ngroups, plo_cob_best_t *out, plo_stats_t *st)
{
    if (!out) return fail(PLO_E_ARG, "null argument");
    { const int rc = cob_check(n, m, TM, Cand, row, offsetblock, coeffs, ncoeffs, p); if (rc != PLO_OK) return rc; }
    if (first_group > (uint64_t)ncoeffs * ncoeffs * ncoeffs || ngroups > (uint64_t)ncoeffs * ncoeffs * ncoeffs - first_group) return fail(PLO_E_ARG, "group range outside the (i,j,k) prefixes");
    if (g_device < 0) { int rc = plo_init(0); if (rc != PLO_OK) return rc; }
    plo_stats_t local{}; if (!st) st = &local; else *st = plo_stats_t{};
    auto t0 = std::chrono::steady_clock::now();
    CobPrep R; cob_prepare(n, m, TM, Cand, row, offsetblock, p, w0, w1, R);
    if (R.dependent) {      // chosen rows are dependent: rank(Cand with row := w) can never exceed `row` (:174)
        out->found = 0; out->zeros_v = w0; out->zeros_w = w1; out->index = 0;
        st->candidates = ngroups * ncoeffs;
        return PLO_OK;
    }
    const uint32_t qn = R.qn, fb = R.fb; const std::vector<uint32_t> &nb = R.nb, &tmb = R.tmb; const unsigned long long init = R.init;
    const uint64_t first = first_group * ncoeffs, total = (first_group + ngroups) * ncoeffs;      // flattened (i,j,k,l) range of this call
    const size_t lds = (4 * (size_t)m + 4 * (size_t)qn + ncoeffs) * 4;
    if (lds > g_lds_max) return fail(PLO_E_CAPACITY, "block of TM does not fit LDS");
    // one device buffer kept by the context, one upload: [best word (2) | TM block | nullspace block | coefficients]
    DevCtx &cx = cur_ctx();
    const size_t o_tm = 2, o_nb = o_tm + tmb.size(), o_cf = o_nb + nb.size(), words = o_cf + ncoeffs;
    if (cx.cob_words < words) {
        if (cx.cob_buf) (void)hipFree(cx.cob_buf);
        cx.cob_buf = nullptr; cx.cob_words = 0;
        HIPCHK(hipMalloc((void **)&cx.cob_buf, (words + 1024) * 4)); cx.cob_words = words + 1024;
    }
    if (!cx.cob_e0) { HIPCHK(hipEventCreate(&cx.cob_e0)); HIPCHK(hipEventCreate(&cx.cob_e1)); }
    std::vector<uint32_t> up(words);
    memcpy(up.data(), &init, 8); memcpy(up.data() + o_tm, tmb.data(), tmb.size() * 4); memcpy(up.data() + o_nb, nb.data(), nb.size() * 4); memcpy(up.data() + o_cf, coeffs, (size_t)ncoeffs * 4);
    HIPCHK(hipMemcpyAsync(cx.cob_buf, up.data(), words * 4, hipMemcpyHostToDevice, g_stream));
    uint32_t *d_tm = cx.cob_buf + o_tm, *d_nb = cx.cob_buf + o_nb, *d_cf = cx.cob_buf + o_cf; unsigned long long *d_best = (unsigned long long *)cx.cob_buf;
    plo::CobJob J{}; J.n = n; J.m = m; J.qn = qn; J.fb = fb; J.C = ncoeffs; J.p = p; J.mu = (~0ull) / p; J.first = first; J.total = total;
    J.tm = d_tm; J.nb = d_nb; J.coeffs = d_cf; J.best = d_best;
    // table form when 4*C*(m+qn) products fit LDS (odd strides: the lanes of a wave read rows l, l+1, ... of one column)
    const uint32_t mstride = m | 1u, qstride = std::max<uint32_t>(qn, 1u) | 1u;
    const size_t lds_tab = 4ull * ncoeffs * ((size_t)mstride + qstride) * 4;
    const bool use_tab = lds_tab <= 72u * 1024u && !getenv("PLO_COB_GENERIC");
    hipEvent_t e0 = cx.cob_e0, e1 = cx.cob_e1;
    uint64_t grid;
    if (use_tab) {
        HIPCHK(hipFuncSetAttribute((const void *)plo::cob_tab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tab));
        grid = std::max<uint64_t>(1, std::min<uint64_t>((ngroups + 3) / 4, (uint64_t)g_cus * 2));
        HIPCHK(hipEventRecord(e0, g_stream));
        hipLaunchKernelGGL(plo::cob_tab_kernel, dim3((uint32_t)grid), dim3(256), lds_tab, g_stream, J, mstride, qstride);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *)plo::cob_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        grid = std::max<uint64_t>(1, std::min<uint64_t>((total - first + 255) / 256, (uint64_t)g_cus * 8));
        HIPCHK(hipEventRecord(e0, g_stream));
        hipLaunchKernelGGL(plo::cob_kernel, dim3((uint32_t)grid), dim3(256), lds, g_stream, J);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, g_stream)); HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long w = 0;
    HIPCHK(hipMemcpy(&w, d_best, 8, hipMemcpyDeviceToHost));
    cob_decode(w, init, n, w0, w1, out);
    st->kernel_ms = ms; st->launches = 1; st->candidates = total - first; st->grid = (uint32_t)grid; st->lds_bytes = (uint32_t)(use_tab ? lds_tab : lds); st->waves_per_wg = 4;
    st->algo_bytes = 16ull * m + 8;                       // the 4 x m block of TM, once, plus the result word
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PLO_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- trilplacer
struct plo_tril_plan {
    plo::TrilPlan P{};
    void *d_img = nullptr; uint32_t *d_err = nullptr; unsigned long long *d_best = nullptr;
    uint32_t waves_per_wg = 4, lds_bytes = 0, blocks_per_cu = 1;
    uint64_t algo_bytes = 0;
    bool rational = false;           // coefficients other than +-1: residues modulo a 31-bit prime, tril_kernel<true>
};

namespace {
int tril_launch(plo_tril_plan *pl, plo::TrilJob J, plo_stats_t *st) {
    HIPCHK(hipMemsetAsync(pl->d_err, 0, sizeof(uint32_t), g_stream));
    J.err = pl->d_err;
    const uint64_t need = (J.ncand + pl->waves_per_wg - 1) / pl->waves_per_wg;
    const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)g_cus * pl->blocks_per_cu, need));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g_stream));
    if (pl->rational) hipLaunchKernelGGL((plo::tril_kernel<true>), dim3((uint32_t)grid), dim3(64 * pl->waves_per_wg), pl->lds_bytes, g_stream, pl->P, J);
    else hipLaunchKernelGGL((plo::tril_kernel<false>), dim3((uint32_t)grid), dim3(64 * pl->waves_per_wg), pl->lds_bytes, g_stream, pl->P, J);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, g_stream)); HIPCHK(hipEventSynchronize(e1));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, pl->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (st) { st->kernel_ms += ms; st->launches += 1; st->grid = (uint32_t)grid; st->lds_bytes = pl->lds_bytes; st->waves_per_wg = pl->waves_per_wg; st->algo_bytes = pl->algo_bytes; st->candidates += J.ncand; }
#ifdef PLO_TRIL_PROFILE
    { unsigned long long gp[8] = {0}; if (hipMemcpyFromSymbol(gp, HIP_SYMBOL(plo::g_tprof), sizeof gp) == hipSuccess && gp[5]) fprintf(stderr, "# tril profile (wave 0 of every workgroup, %llu candidates): cycles per candidate: perm %.0f build %.0f pushvariables %.0f simplify %.0f; fixpoint trips %.1f\n", gp[5], (double)gp[0] / gp[5], (double)gp[1] / gp[5], (double)gp[2] / gp[5], (double)gp[3] / gp[5], (double)gp[4] / gp[5]); }
#endif
    if (err) return fail(err == plo::TERR_CAP ? PLO_E_INTERNAL : PLO_E_UNSUPPORTED, "device (trilplacer): error " + std::to_string(err));
    return PLO_OK;
}
} // namespace

extern "C" {

int plo_tril_plan_create(const plo_icsr_t *A, const plo_icsr_t *B, const plo_icsr_t *T, plo_tril_plan_t **plan) { return plo_tril_plan_create_x(A, B, T, 0, plan); }

// Common builder: entries as rationals num/den.  All entries +-1: the unit kernel (small signed values); otherwise the residues
// modulo PLO_TRIL_PRIME and the rational instantiation of the kernel (plo_tril.hip).
#define PLO_TRIL_PRIME 2147483629u
int plo_tril_plan_create_q(const plo_qcsr_t *A, const plo_qcsr_t *B, const plo_qcsr_t *T, int expanded, plo_tril_plan_t **plan)
{
    if (g_device < 0) return fail(PLO_E_HIP, "plo_init was not called (or found no HIP device)");
    if (!A || !B || !T || !plan) return fail(PLO_E_ARG, "null argument");
    const plo_qcsr_t *Ms[3] = {A, B, T};
    if (A->m == 0 || A->m != B->m || A->m != T->m) return fail(PLO_E_ARG, "A, B and T (transposed product matrix) need the same number of rows");
    if (A->m > 16382u) return fail(PLO_E_CAPACITY, "more than 16382 rows");
    uint32_t cap = 0; size_t bytes = 0; uint64_t algo = 8; bool unit = true;
    for (const plo_qcsr_t *M : Ms) {
        if (!M->rowptr || !M->col || !M->num || M->n == 0 || M->n > 16382u) return fail(PLO_E_ARG, "bad matrix (at most 16382 variables)");
        const uint32_t nnz = M->rowptr[M->m];
        if (nnz > 65535u) return fail(PLO_E_CAPACITY, "more than 65535 non-zeros");
        for (uint32_t i = 0; i < M->m; ++i) {
            const uint32_t len = M->rowptr[i + 1] - M->rowptr[i];
            if (len == 0) return fail(PLO_E_UNSUPPORTED, "empty row: host path only");
            if (len > 64) return fail(PLO_E_UNSUPPORTED, "row with more than 64 entries: host path only");
            for (uint32_t e = M->rowptr[i]; e < M->rowptr[i + 1]; ++e) {
                const int64_t nu = M->num[e], de = M->den ? M->den[e] : 1;
                if (nu == 0 || de == 0) return fail(PLO_E_ARG, "zero entry or zero denominator");
                if (!(de == 1 && (nu == 1 || nu == -1))) unit = false;
                if (de % (int64_t)PLO_TRIL_PRIME == 0 || nu % (int64_t)PLO_TRIL_PRIME == 0) return fail(PLO_E_UNSUPPORTED, "entry not a unit modulo the device's prime: host path only");
                if (M->col[e] >= M->n || (e > M->rowptr[i] && M->col[e] <= M->col[e - 1])) return fail(PLO_E_ARG, "columns must be sorted and in range");
            }
        }
        cap = std::max(cap, 2u * nnz + 3u * M->m);
        if (expanded && M == T) cap = std::max(cap, 4u * nnz + 6u * M->m);        // TransposedDoubleAlgorithm: 4(len-1)+2 atoms per row, and 4 scaling atoms when the pivot is not +-1
        if (expanded && M == T && M->n >= 16382u) return fail(PLO_E_CAPACITY, "one more variable of c than the atom holds");
        bytes += round_up((M->m + 1) * 2, 16) + round_up(nnz * 2, 16) + round_up(nnz, 16) + round_up(nnz * 4, 16);
        algo += 2ull * (M->m + 1) + 3ull * nnz;                 // the CSR image of the three matrices, once per candidate
    }
    cap = round_up(cap + 2, 64);
    plo_tril_plan *pl = new plo_tril_plan();
    pl->rational = !unit;
    std::vector<uint8_t> img(bytes, 0);
    if (hipMalloc(&pl->d_img, bytes) != hipSuccess) { delete pl; return fail(PLO_E_HIP, "hipMalloc"); }
    size_t off = 0;
    for (int w = 0; w < 3; ++w) {
        const plo_qcsr_t *M = Ms[w]; const uint32_t nnz = M->rowptr[M->m];
        plo::TrilMat &D = pl->P.M[w];
        D.m = M->m; D.n = M->n; D.nnz = nnz;
        uint16_t *rp = (uint16_t *)(img.data() + off); D.rp = (const uint16_t *)((uint8_t *)pl->d_img + off); off += round_up((M->m + 1) * 2, 16);
        uint16_t *cl = (uint16_t *)(img.data() + off); D.col = (const uint16_t *)((uint8_t *)pl->d_img + off); off += round_up(nnz * 2, 16);
        int8_t *vl = (int8_t *)(img.data() + off); D.val = (const int8_t *)((uint8_t *)pl->d_img + off); off += round_up(nnz, 16);
        uint32_t *vp = (uint32_t *)(img.data() + off); D.valp = (const uint32_t *)((uint8_t *)pl->d_img + off); off += round_up(nnz * 4, 16);
        for (uint32_t i = 0; i <= M->m; ++i) rp[i] = (uint16_t)M->rowptr[i];
        for (uint32_t e = 0; e < nnz; ++e) {
            cl[e] = (uint16_t)M->col[e];
            const int64_t nu = M->num[e], de = M->den ? M->den[e] : 1;
            vl[e] = unit ? (int8_t)nu : 0;
            int64_t a = nu % (int64_t)PLO_TRIL_PRIME, d = de % (int64_t)PLO_TRIL_PRIME; if (a < 0) a += PLO_TRIL_PRIME; if (d < 0) d += PLO_TRIL_PRIME;
            vp[e] = (uint32_t)((uint64_t)a * inv_mod((uint32_t)d, PLO_TRIL_PRIME) % PLO_TRIL_PRIME);
        }
    }
    pl->P.cap = cap; pl->P.expanded = expanded ? 1u : 0u; pl->P.p = unit ? 0u : PLO_TRIL_PRIME;
    pl->P.lds_per_wave = round_up(8u * cap + 2u * ((A->m + 1u) & ~1u) + A->m, 16) + 16u * ((cap + 63u) / 64u);   // atoms, permutation, signs; masks of a pushvariables pass
    pl->algo_bytes = algo;
    pl->waves_per_wg = 4;
    pl->lds_bytes = pl->P.lds_per_wave * pl->waves_per_wg;
    if (pl->lds_bytes > 64u * 1024u) { pl->waves_per_wg = 1; pl->lds_bytes = pl->P.lds_per_wave; }
    if (pl->lds_bytes > g_lds_max) { (void)hipFree(pl->d_img); delete pl; return fail(PLO_E_CAPACITY, "program does not fit LDS"); }
    pl->blocks_per_cu = std::max<uint32_t>(1, std::min<uint32_t>(32u / pl->waves_per_wg, (uint32_t)(g_lds_max / pl->lds_bytes)));
    const void *fn = pl->rational ? (const void *)plo::tril_kernel<true> : (const void *)plo::tril_kernel<false>;
    if (hipMemcpy(pl->d_img, img.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMalloc((void **)&pl->d_err, 4) != hipSuccess ||
        hipMalloc((void **)&pl->d_best, 8) != hipSuccess || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes) != hipSuccess) {
        plo_tril_plan_destroy(pl); return fail(PLO_E_HIP, "device setup of the trilplacer plan failed");
    }
    *plan = pl;
    return PLO_OK;
}

// integer entries (the interface of rounds 1-2): the same builder with denominators 1
int plo_tril_plan_create_x(const plo_icsr_t *A, const plo_icsr_t *B, const plo_icsr_t *T, int expanded, plo_tril_plan_t **plan)
{
    if (!A || !B || !T || !plan) return fail(PLO_E_ARG, "null argument");
    const plo_icsr_t *Ms[3] = {A, B, T};
    std::vector<int64_t> nums[3]; plo_qcsr_t Q[3];
    for (int w = 0; w < 3; ++w) {
        if (!Ms[w]->rowptr || !Ms[w]->col || !Ms[w]->val) return fail(PLO_E_ARG, "bad matrix");
        const uint32_t nnz = Ms[w]->rowptr[Ms[w]->m];
        nums[w].assign(Ms[w]->val, Ms[w]->val + nnz);
        Q[w] = plo_qcsr_t{Ms[w]->m, Ms[w]->n, Ms[w]->rowptr, Ms[w]->col, nums[w].data(), nullptr};
    }
    return plo_tril_plan_create_q(&Q[0], &Q[1], &Q[2], expanded, plan);
}

void plo_tril_plan_destroy(plo_tril_plan_t *pl)
{
    if (!pl) return;
    if (pl->d_img) (void)hipFree(pl->d_img);
    if (pl->d_err) (void)hipFree(pl->d_err);
    if (pl->d_best) (void)hipFree(pl->d_best);
    delete pl;
}

int plo_tril_cost_many(plo_tril_plan_t *pl, const uint64_t *seeds, uint64_t seed0, uint64_t n, uint32_t *ops6, plo_stats_t *stats)
{
    if (!pl || !ops6) return fail(PLO_E_ARG, "null argument");
    if (n >= (1ull << 31)) return fail(PLO_E_ARG, "at most 2^31-1 candidates per call");
    plo_stats_t local{}; plo_stats_t *st = stats ? stats : &local; *st = plo_stats_t{};
    const auto t0 = std::chrono::steady_clock::now();
    if (n == 0) return PLO_OK;
    uint32_t *d_ops = nullptr; uint64_t *d_seeds = nullptr;
    HIPCHK(hipMalloc((void **)&d_ops, n * 6 * sizeof(uint32_t)));
    if (seeds) { HIPCHK(hipMalloc((void **)&d_seeds, n * 8)); HIPCHK(hipMemcpy(d_seeds, seeds, n * 8, hipMemcpyHostToDevice)); }
    plo::TrilJob J{}; J.seed0 = seed0; J.seeds = d_seeds; J.ncand = n; J.ops = d_ops; J.best = nullptr;
    int rc = tril_launch(pl, J, st);
    if (rc == PLO_OK && hipMemcpy(ops6, d_ops, n * 6 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(PLO_E_HIP, "copy back");
    (void)hipFree(d_ops); if (d_seeds) (void)hipFree(d_seeds);
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int plo_tril_search(plo_tril_plan_t *pl, uint64_t seed0, uint64_t nseeds, plo_tril_best_t *best, plo_stats_t *stats)
{
    if (!pl || !best) return fail(PLO_E_ARG, "null argument");
    if (nseeds == 0 || nseeds >= (1ull << 31)) return fail(PLO_E_ARG, "1 .. 2^31-1 candidates per call");
    plo_stats_t local{}; plo_stats_t *st = stats ? stats : &local; *st = plo_stats_t{};
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long init = ~0ull;
    HIPCHK(hipMemcpy(pl->d_best, &init, 8, hipMemcpyHostToDevice));
    plo::TrilJob J{}; J.seed0 = seed0; J.seeds = nullptr; J.ncand = nseeds; J.ops = nullptr; J.best = pl->d_best;
    int rc = tril_launch(pl, J, st);
    if (rc != PLO_OK) return rc;
    unsigned long long w = 0;
    HIPCHK(hipMemcpy(&w, pl->d_best, 8, hipMemcpyDeviceToHost));
    if (w == init) return fail(PLO_E_INTERNAL, "no candidate reported");
    best->add = (uint32_t)(w >> 48); best->sca = (uint32_t)(w >> 32) & 0xFFFFu; best->mul = pl->P.M[0].m;
    best->variant = (uint32_t)(w & 1ull); best->seed = seed0 + ((w & 0xFFFFFFFFull) >> 1);
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return PLO_OK;
}

int plo_kernel_search_multi(const plo_csr_t *M, uint32_t p, uint64_t seed0, uint64_t nrestarts, uint32_t per_block, int cost_mode,
                            int ndev, const int *devices, plo_best_t *out, plo_stats_t *stats)
{
    if (!M || !out) return fail(PLO_E_ARG, "null argument");
    if (ndev < 1 || ndev > 64) return fail(PLO_E_ARG, "device count outside [1,64]");
    if (cost_mode < 0 || cost_mode > 2) return fail(PLO_E_ARG, "unknown cost mode");
    if (per_block != 1u && ndev > 1) return fail(PLO_E_ARG, "shards of the kernel method take one decomposition per restart (per_block = 1)");
    DeviceGuard guard;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<MultiShard> sh; std::vector<plo_best_t> bests((size_t)ndev);
    auto key = [&](const plo_best_t &b) -> unsigned long long {
        switch (cost_mode) { case PLO_COST_ADD_THEN_MUL: return ((unsigned long long)b.adds << 32) | b.muls; case PLO_COST_SUM: return ((unsigned long long)b.adds + b.muls) << 32; default: return (((unsigned long long)b.adds + b.muls) << 32) | b.adds; } };
    int rc = multi_run(ndev, devices, seed0, nrestarts, sh, [&](MultiShard &S, int) {
        plo_best_t &b = bests[(size_t)(&S - sh.data())];
        const int r_ = plo_kernel_search(M, p, S.s0, S.cnt, per_block, cost_mode, nullptr, nullptr, nullptr, &b, &S.st);
        if (r_ == PLO_OK && b.seed != ~0ull) { S.hi = key(b); S.lo = b.seed - seed0; }
        return r_;
    });
    if (rc != PLO_OK) return rc;
    plo_stats_t agg{}; int win = -1;
    rc = multi_min(sh, ndev, devices, agg, win);
    if (rc != PLO_OK) return rc;
    out->adds = out->muls = 0xFFFFFFFFu; out->seed = ~0ull;
    if (win >= 0) *out = bests[(size_t)win];
    agg.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = agg;
    return PLO_OK;
}

int plo_tril_search_multi(const plo_qcsr_t *A, const plo_qcsr_t *B, const plo_qcsr_t *T, int expanded, uint64_t seed0, uint64_t nseeds,
                          int ndev, const int *devices, plo_tril_best_t *best, plo_stats_t *stats)
{
    if (!A || !B || !T || !best) return fail(PLO_E_ARG, "null argument");
    if (ndev < 1 || ndev > 64) return fail(PLO_E_ARG, "device count outside [1,64]");
    if (nseeds == 0 || nseeds >= (1ull << 62)) return fail(PLO_E_ARG, "1 .. 2^62-1 candidates per call");
    DeviceGuard guard;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<MultiShard> sh; std::vector<plo_tril_best_t> bests((size_t)ndev);
    int rc = multi_run(ndev, devices, seed0, nseeds, sh, [&](MultiShard &S, int) {
        plo_tril_best_t &b = bests[(size_t)(&S - sh.data())];
        plo_tril_plan_t *plan = nullptr;
        int r_ = plo_tril_plan_create_q(A, B, T, expanded, &plan);
        if (r_ != PLO_OK) return r_;
        // (a launch takes at most 2^31-1 candidates: longer shards go in pieces, minimum under the same order)
        bool have = false;
        for (uint64_t done = 0; r_ == PLO_OK && done < S.cnt;) {
            const uint64_t piece = std::min<uint64_t>(S.cnt - done, (1ull << 31) - 1ull);
            plo_tril_best_t pb{}; plo_stats_t ps{};
            r_ = plo_tril_search(plan, S.s0 + done, piece, &pb, &ps);
            if (r_ != PLO_OK) break;
            S.st.candidates += ps.candidates; S.st.launches += ps.launches; S.st.kernel_ms += ps.kernel_ms; S.st.grid = ps.grid; S.st.lds_bytes = ps.lds_bytes; S.st.waves_per_wg = ps.waves_per_wg; S.st.algo_bytes = ps.algo_bytes;
            const unsigned long long hi = ((unsigned long long)pb.add << 32) | pb.sca, lo = ((pb.seed - seed0) << 1) | (pb.variant & 1u);      // the order of include/plinopt_inplace.inl:893-897, then (seed, variant)
            if (!have || hi < S.hi || (hi == S.hi && lo < S.lo)) { S.hi = hi; S.lo = lo; b = pb; have = true; }
            done += piece;
        }
        plo_tril_plan_destroy(plan);
        return r_;
    });
    if (rc != PLO_OK) return rc;
    plo_stats_t agg{}; int win = -1;
    rc = multi_min(sh, ndev, devices, agg, win);
    if (rc != PLO_OK) return rc;
    if (win < 0) return fail(PLO_E_INTERNAL, "no candidate reported");
    *best = bests[(size_t)win];
    agg.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = agg;
    return PLO_OK;
}

} // extern "C"
