// ===========================================================================
// plo_cob.hip -- gfx950 kernel for the change-of-basis search of bin/sparsifier:
// one (block,row) enumeration of `localSparsifier` (reference
// include/plinopt_sparsify.inl:282-314), i.e. |Coeffs|^4 evaluations of
// `testLinComb` (:167-197), one candidate row w per lane:
//   * w is independent of the rows already chosen  <=>  w . N != 0 for a basis N of
//     their right nullspace (computed once on the host; the reference copies the
//     candidate matrix and runs a Gaussian elimination per candidate, :38-45,:172-175);
//   * v = TM^T w (only the <= 4 block rows of TM matter), score = (zeros(v), zeros(w));
//   * strictly-better-replaces in lexicographic (i,j,k,l) order == argmax with the
//     smallest index: packed (score, ~index) and one 64-bit atomicMax per wave.
// HBM traffic is the 4 x m block of TM (staged in LDS) and one 8-byte result:
// the kernel is bound by integer VALU issue (4 modular products per column).
// ===========================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

struct CobJob {
    uint32_t n, m, qn, fb, C, p; uint64_t mu; uint64_t first, total;     // candidates first .. total-1 of the C^4 (a shard of the enumeration; whole: 0, C^4)
    const uint32_t *tm;       // 4 x m   block rows of TM (rows beyond n are zero)
    const uint32_t *nb;       // 4 x qn  block rows of the nullspace basis of the chosen rows
    const uint32_t *coeffs;   // C
    unsigned long long *best; // packed (score+1)<<32 | ~index, initialised with the incoming weight
};

__device__ __forceinline__ uint32_t cob_mul(uint32_t a, uint32_t b, uint32_t p, uint64_t mu) {
    uint64_t x = (uint64_t)a * b, q = __umul64hi(x, mu), r = x - q * p;
    while (r >= p) r -= p;
    return (uint32_t)r;
}
__device__ __forceinline__ bool cob_zero4(uint64_t s, uint32_t p) { return s == 0 || s == p || s == 2ull * p || s == 3ull * p; }

__device__ __forceinline__ void cob_body(const CobJob &J)
{
    extern __shared__ uint32_t cl[];                    // tm block (4*m) then nullspace block (4*qn) then coeffs (C)
    uint32_t *tm = cl, *nb = cl + 4u * J.m, *cf = nb + 4u * J.qn;
    for (uint32_t i = threadIdx.x; i < 4u * J.m; i += blockDim.x) tm[i] = J.tm[i];
    for (uint32_t i = threadIdx.x; i < 4u * J.qn; i += blockDim.x) nb[i] = J.nb[i];
    for (uint32_t i = threadIdx.x; i < J.C; i += blockDim.x) cf[i] = J.coeffs[i];
    __syncthreads();
    const uint32_t p = J.p, C = J.C, m = J.m, qn = J.qn, fb = J.fb;
    const uint64_t mu = J.mu;
    uint64_t mybest = 0;
    for (uint64_t idx = J.first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < J.total; idx += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)idx, w[4];
        w[3] = cf[x % C]; x /= C; w[2] = cf[x % C]; x /= C; w[1] = cf[x % C]; x /= C; w[0] = cf[x];
#pragma unroll
        for (int t = 0; t < 4; ++t) if ((uint32_t)t >= fb) w[t] = 0;                 // positions beyond the matrix are dropped (:310)
        bool indep = false;
        for (uint32_t c = 0; c < qn && !indep; ++c) {
            uint64_t s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s += cob_mul(w[t], nb[t * qn + c], p, mu);
            indep = !cob_zero4(s, p);
        }
        if (!indep) continue;
        uint32_t zv = 0, zw = J.n - fb;
#pragma unroll
        for (int t = 0; t < 4; ++t) if ((uint32_t)t < fb && w[t] == 0) ++zw;
        for (uint32_t c = 0; c < m; ++c) {
            uint64_t s = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s += cob_mul(w[t], tm[t * m + c], p, mu);
            zv += cob_zero4(s, p) ? 1u : 0u;
        }
        const uint64_t key = ((uint64_t)(zv * (J.n + 1u) + zw + 1u) << 32) | (uint32_t)(~(uint32_t)idx);
        mybest = key > mybest ? key : mybest;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mybest, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mybest >> 32), o);
        const uint64_t v = ((uint64_t)hi << 32) | lo; mybest = v > mybest ? v : mybest;
    }
    if ((threadIdx.x & 63u) == 0 && mybest) atomicMax(J.best, (unsigned long long)mybest);
}

// Table form of the same enumeration (used whenever the tables fit LDS).  With P[t][x][c] = coeffs[x] * TM[off+t][c]
// mod p tabulated once per workgroup, v_c(i,j,k,l) = P0[i][c] + P1[j][c] + P2[k][c] + P3[l][c], and v_c == 0 iff
// P3[l][c] == neg_c, where neg_c = -(P0[i][c]+P1[j][c]+P2[k][c]) mod p depends on (i,j,k) only.  One wavefront
// takes one (i,j,k); its lanes take l; per column the wave does three broadcast LDS reads and a reduction for
// neg_c, each lane one LDS read and one compare.  The independence test w.N != 0 has the same shape on the
// tabulated products with the nullspace block.  ~0.2 wave instructions per candidate and column instead of ~90.
__device__ __forceinline__ void cob_tab_body(const CobJob &J, uint32_t ms /* odd row stride >= m */, uint32_t qs /* odd stride >= qn */)
{
    extern __shared__ uint32_t cl[];
    const uint32_t p = J.p, C = J.C, m = J.m, qn = J.qn, fb = J.fb;
    const uint64_t mu = J.mu;
    uint32_t *PT = cl;                       // [4][C][ms]
    uint32_t *PN = cl + 4u * C * ms;         // [4][C][qs]
    for (uint32_t x = threadIdx.x; x < 4u * C * ms; x += blockDim.x) {
        const uint32_t c = x % ms, ci = (x / ms) % C, t = x / (ms * C);
        PT[x] = (c < m && t < fb) ? cob_mul(J.coeffs[ci], J.tm[t * m + c], p, mu) : 0u;
    }
    for (uint32_t x = threadIdx.x; x < 4u * C * qs; x += blockDim.x) {
        const uint32_t c = x % qs, ci = (x / qs) % C, t = x / (qs * C);
        PN[x] = (c < qn && t < fb) ? cob_mul(J.coeffs[ci], J.nb[t * qn + c], p, mu) : 0u;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, nwaves = blockDim.x >> 6;
    const uint64_t ngroups = J.total / C;                     // groups (i,j,k) J.first/C .. J.total/C - 1: shards are cut at group boundaries
    uint64_t mybest = 0;
    for (uint64_t gidx = J.first / C + (uint64_t)blockIdx.x * nwaves + (threadIdx.x >> 6); gidx < ngroups; gidx += (uint64_t)gridDim.x * nwaves) {
        uint32_t x = (uint32_t)gidx; const uint32_t k = x % C; x /= C; const uint32_t j = x % C, i = x / C;
        const uint32_t *p0 = PT + (0u * C + i) * ms, *p1 = PT + (1u * C + j) * ms, *p2 = PT + (2u * C + k) * ms;
        const uint32_t *n0 = PN + (0u * C + i) * qs, *n1 = PN + (1u * C + j) * qs, *n2 = PN + (2u * C + k) * qs;
        uint32_t zfix = J.n - fb;
        if (fb > 0u && J.coeffs[i] == 0u) ++zfix;
        if (fb > 1u && J.coeffs[j] == 0u) ++zfix;
        if (fb > 2u && J.coeffs[k] == 0u) ++zfix;
        for (uint32_t l0 = 0; l0 < C; l0 += 64u) {
            const uint32_t l = l0 + lane; const bool act = l < C;
            const uint32_t *p3 = PT + (3u * C + (act ? l : 0u)) * ms, *n3 = PN + (3u * C + (act ? l : 0u)) * qs;
            bool indep = false;
            for (uint32_t c = 0; c < qn; ++c) {
                uint64_t s = (uint64_t)n0[c] + n1[c] + n2[c];
                if (s >= p) s -= p; if (s >= p) s -= p;
                const uint32_t neg = s ? p - (uint32_t)s : 0u;
                indep |= n3[c] != neg;
            }
            uint32_t zv = 0;
            for (uint32_t c = 0; c < m; ++c) {
                uint64_t s = (uint64_t)p0[c] + p1[c] + p2[c];
                if (s >= p) s -= p; if (s >= p) s -= p;
                const uint32_t neg = s ? p - (uint32_t)s : 0u;
                zv += (p3[c] == neg) ? 1u : 0u;
            }
            if (act && indep) {
                const uint32_t zw = zfix + ((fb > 3u && J.coeffs[l] == 0u) ? 1u : 0u);
                const uint64_t idx = gidx * C + l;
                const uint64_t key = ((uint64_t)(zv * (J.n + 1u) + zw + 1u) << 32) | (uint32_t)(~(uint32_t)idx);
                mybest = key > mybest ? key : mybest;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mybest, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mybest >> 32), o);
        const uint64_t v = ((uint64_t)hi << 32) | lo; mybest = v > mybest ? v : mybest;
    }
    if (lane == 0 && mybest) atomicMax(J.best, (unsigned long long)mybest);
}

__global__ __launch_bounds__(256) void cob_kernel(CobJob J) { cob_body(J); }
__global__ __launch_bounds__(256) void cob_tab_kernel(CobJob J, uint32_t ms, uint32_t qs) { cob_tab_body(J, ms, qs); }

// Several enumerations of the same shape in ONE launch (blockIdx.y = the enumeration): the two primes of an enumeration over the
// rationals (CobGpuQBackend of bin/sparsifier), or the shards of one.  Table form when `tab`.
#define PLO_COB_BATCH 4
struct CobBatch { CobJob J[PLO_COB_BATCH]; uint32_t ms[PLO_COB_BATCH], qs[PLO_COB_BATCH]; uint32_t tab; };
__global__ __launch_bounds__(256) void cob_batch_kernel(CobBatch B)
{
    const uint32_t y = blockIdx.y;
    // (a switch on the uniform index: the jobs stay in the kernel-argument segment instead of being copied to scratch)
    if (B.tab) { if (y == 0u) cob_tab_body(B.J[0], B.ms[0], B.qs[0]); else if (y == 1u) cob_tab_body(B.J[1], B.ms[1], B.qs[1]); else if (y == 2u) cob_tab_body(B.J[2], B.ms[2], B.qs[2]); else cob_tab_body(B.J[3], B.ms[3], B.qs[3]); }
    else { if (y == 0u) cob_body(B.J[0]); else if (y == 1u) cob_body(B.J[1]); else if (y == 2u) cob_body(B.J[2]); else cob_body(B.J[3]); }
}

} // namespace plo
