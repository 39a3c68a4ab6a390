// ==========================================================================
// plo_tril.hip -- in-place trilinear program search (trilplacer) on gfx950.
//
// Replaces the body of the restart loop of SearchTriLinearAlgorithm
// (reference include/plinopt_inplace.inl:837-924): one candidate = a row permutation
// and coherent row negations of (A, B, T), then the oriented and the unoriented
// in-place program (LinearAlgorithm :400-502 on A, B and, transposed, on T) and
// their (ADD, SCA) counts (complexity :133-144).  Counts only: the host replays the
// winning seed to print the program.
//
// One wavefront per candidate.  The atom list of the program being built lives in
// LDS as four arrays (src u16, des i16, val i16, ope u8); the control flow is wave
// uniform and the lanes share the scans:
//   simplify  (:243-311)  64 atoms at a time, every lane walks the atoms after its
//                         own one until it meets a mergeable atom or a dependency;
//                         the first lane (program order) that found a merge applies it;
//   pushvariables (:322-393)  per output variable, "next atom of the variable" and
//                         "first atom that stops the push" are ballot scans; a rotation
//                         is a parallel shift of the LDS arrays.
// Coefficients: device path for matrices with entries +-1 and no empty row (every
// multiplicative atom is then *-1 and additive values are small integers); anything
// else is reported as unsupported and stays on the host.
// ==========================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

struct TrilMat { uint32_t m, n, nnz; const uint16_t *rp; const uint16_t *col; const int8_t *val; };
struct TrilPlan { TrilMat M[3]; uint32_t cap; uint32_t lds_per_wave; };
struct TrilJob {
    uint64_t seed0; const uint64_t *seeds; uint64_t ncand;
    uint32_t *ops;               // 6 per candidate: ADD,SCA,MUL oriented then unoriented (may be null)
    unsigned long long *best;    // packed minimum (may be null)
    uint32_t *err;
};
enum { TERR_CAP = 21, TERR_ROW = 22 };
enum { T_BAR = 0, T_ADD = 1, T_SUB = 2, T_MUL = 3, T_DIV = 4 };

#ifndef PLO_TRIL_LOCKSTEPS
#define PLO_TRIL_LOCKSTEPS 6u
#endif
#define TW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

#ifdef PLO_TRIL_PROFILE
__device__ unsigned long long g_tprof[8];   // lane 0 of wave 0 of every workgroup: cycles in perm, build, pushvariables, simplify; calls
#define TP_ADD(k_, t0_) do { if (threadIdx.x == 0) g_tprof[k_] += clock64() - (t0_); } while (0)
#else
#define TP_ADD(k_, t0_) do { } while (0)
#endif
struct TrilProg { uint16_t *src; int16_t *des; int16_t *val; uint8_t *ope; uint32_t n; };

__device__ __forceinline__ uint32_t t_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t t_rng(uint32_t &s) { s = (uint32_t)((950706376ull * (uint64_t)s) % 2147483647ull); return s; }
__device__ __forceinline__ uint64_t t_splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__device__ __forceinline__ bool t_as(uint32_t o) { return o == T_ADD || o == T_SUB; }
__device__ __forceinline__ bool t_md(uint32_t o) { return o == T_MUL || o == T_DIV; }

// remove atoms [k, k+1): everything behind moves one slot down (64 at a time, lowest chunk first: a chunk's reads are
// complete before its writes, and it reads the first element of the next chunk before that chunk is written)
__device__ __forceinline__ void t_erase(TrilProg &P, uint32_t k, uint32_t lane) {
    for (uint32_t b = k; b + 1u < P.n; b += 64u) {
        const uint32_t d = b + lane; const bool on = d + 1u < P.n;
        uint16_t s = 0; int16_t e = 0, v = 0; uint8_t o = 0;
        if (on) { s = P.src[d + 1u]; e = P.des[d + 1u]; v = P.val[d + 1u]; o = P.ope[d + 1u]; }
        TW_SYNC();
        if (on) { P.src[d] = s; P.des[d] = e; P.val[d] = v; P.ope[d] = o; }
        TW_SYNC();
    }
    --P.n;
}
// std::rotate(f, f+1, e): atom f goes to e-1, [f+1, e) moves one slot down
__device__ __forceinline__ void t_rotate(TrilProg &P, uint32_t f, uint32_t e, uint32_t lane) {
    const uint16_t fs = P.src[f]; const int16_t fd = P.des[f], fv = P.val[f]; const uint8_t fo = P.ope[f];
    TW_SYNC();
    for (uint32_t b = f; b + 1u < e; b += 64u) {
        const uint32_t d = b + lane; const bool on = d + 1u < e;
        uint16_t s = 0; int16_t x = 0, v = 0; uint8_t o = 0;
        if (on) { s = P.src[d + 1u]; x = P.des[d + 1u]; v = P.val[d + 1u]; o = P.ope[d + 1u]; }
        TW_SYNC();
        if (on) { P.src[d] = s; P.des[d] = x; P.val[d] = v; P.ope[d] = o; }
        TW_SYNC();
    }
    if (lane == 0) { P.src[e - 1u] = fs; P.des[e - 1u] = fd; P.val[e - 1u] = fv; P.ope[e - 1u] = fo; }
    TW_SYNC();
}

// first k in [start, n) with pred(k), or n
template <class F> __device__ __forceinline__ uint32_t t_find(uint32_t start, uint32_t n, uint32_t lane, F pred) {
    for (uint32_t b = start; b < n; b += 64u) {
        const uint32_t k = b + lane;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(k < n && pred(k));
        if (m) return b + (uint32_t)__builtin_ctzll(m);
    }
    return n;
}

// :243-311.  true = one merge applied.
__device__ bool t_simplify(TrilProg &P, bool transposed, uint32_t lane) {
    for (uint32_t base = 0; base < P.n; base += 64u) {
        const uint32_t p = base + lane, n = P.n;
        bool act = p < n;
        int is = 0, id = 0, iv = 0; uint32_t io = 0;
        if (act) { is = P.src[p]; id = P.des[p]; iv = P.val[p]; io = P.ope[p]; act = io != T_BAR; }
        uint32_t k = p + 1u, q = 0; int res = act ? 0 : 2;       // 0 walking, 1 merge found, 2 stopped
        // what stops the walk of atom (is,id,io) at atom kk: a merge (sameops + cumulate) or a dependency (:284-307)
        auto merges = [&](int s_, int d_, uint32_t o_, uint32_t kk) {
            const uint32_t no = P.ope[kk];
            return (int)P.src[kk] == s_ && (int)P.des[kk] == d_ && ((t_as(o_) && t_as(no)) || (t_md(o_) && t_md(no)));
        };
        auto breaks = [&](int s_, int d_, uint32_t o_, uint32_t kk) {
            const int ns = P.src[kk], nd = P.des[kk]; const uint32_t no = P.ope[kk];
            bool brk = (s_ == ns) && (no == T_BAR || (t_as(o_) && t_md(no)) || (t_md(o_) && t_as(no)));
            brk |= transposed ? (d_ == ns) : (d_ == ns && no != T_BAR);
            brk |= (s_ == nd);
            return brk;
        };
        // most walks end within a few atoms: a few lock-step steps, one atom per lane ...
        for (uint32_t step = 0; step < PLO_TRIL_LOCKSTEPS && __builtin_amdgcn_ballot_w64(res == 0); ++step) {
            if (res == 0) {
                if (k >= n) res = 2;
                else if (merges(is, id, io, k)) { res = 1; q = k; }
                else if (breaks(is, id, io, k)) res = 2;
                else ++k;
            }
        }
        // ... and each long walk that is still open before the first merge found so far is finished by the whole wave
        {
            const unsigned long long ok1 = __builtin_amdgcn_ballot_w64(res == 1);
            unsigned long long und = __builtin_amdgcn_ballot_w64(res == 0);
            if (ok1) und &= (1ull << __builtin_ctzll(ok1)) - 1ull;
            while (und) {
                const int L = __builtin_ctzll(und); und &= und - 1ull;
                const int bs = __builtin_amdgcn_readlane(is, L), bd = __builtin_amdgcn_readlane(id, L);
                const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)io, L), bk = (uint32_t)__builtin_amdgcn_readlane((int)k, L);
                const uint32_t e = t_find(bk, n, lane, [&](uint32_t kk) { return merges(bs, bd, bo, kk) || breaks(bs, bd, bo, kk); });
                const bool mg = e < n && merges(bs, bd, bo, e);
                if ((int)lane == L) { if (mg) { res = 1; q = e; } else res = 2; }
                if (mg) break;
            }
        }
        const unsigned long long ok = __builtin_amdgcn_ballot_w64(res == 1);
        if (ok) {
            const int L = __builtin_ctzll(ok);
            const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)p, L), qq = (uint32_t)__builtin_amdgcn_readlane((int)q, L);
            const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)io, L); const int v1 = __builtin_amdgcn_readlane(iv, L);
            const uint32_t o2 = P.ope[qq]; const int v2 = P.val[qq];
            uint32_t o = o1; int v; bool noop;
            if (t_as(o1)) {                                   // :96-107
                v = (o1 == o2) ? v1 + v2 : v1 - v2;
                if (v < 0) { o = (o1 == T_ADD) ? T_SUB : T_ADD; v = -v; }
                noop = v == 0;
            } else {                                          // :108-118, values are +-1
                v = v1 * v2; noop = v == 1;
            }
            TW_SYNC();
            t_erase(P, qq, lane);
            if (noop) t_erase(P, pp, lane);
            else if (lane == 0) { P.ope[pp] = (uint8_t)o; P.val[pp] = (int16_t)v; }
            TW_SYNC();
            return true;
        }
    }
    return false;
}

// :322-393
__device__ void t_pushvariables(TrilProg &P, uint32_t numout, uint32_t lane) {
    for (uint32_t i = 0; i < numout; ++i) {
        uint32_t pos = 0;
        for (;;) {
            const uint32_t n = P.n;
            const uint32_t f = t_find(pos, n, lane, [&](uint32_t k) { return P.ope[k] != T_BAR && P.src[k] == i; });
            if (f >= n) break;
            const uint32_t fo = P.ope[f]; const int fd = P.des[f];
            uint32_t e;
            if (t_as(fo)) e = t_find(f + 1u, n, lane, [&](uint32_t k) { const int s = P.src[k]; return fd == s || (s == (int)i && (fd == (int)P.des[k] || t_md(P.ope[k]))); });
            else e = t_find(f + 1u, n, lane, [&](uint32_t k) { return (int)P.des[k] == (int)i || P.src[k] == i; });
            if (e >= n) { if (f + 1u != n) t_rotate(P, f, n, lane); break; }      // can be moved to the end (:381-391)
            bool rot;
            if (t_as(fo)) rot = !(fd == (int)P.src[e]) && fd == (int)P.des[e];
            else rot = !((int)P.des[e] == (int)i) && t_md(P.ope[e]);
            if (rot && f + 1u != e) t_rotate(P, f, e, lane);
            pos = e + 1u;
        }
    }
}

// :400-502 for +-1 matrices without empty rows; perm/sign describe the candidate's rows
__device__ void t_linear(TrilProg &P, const TrilMat &M, const uint16_t *perm, const uint8_t *sgn, uint32_t sbit, bool transposed,
                         bool oriented, uint32_t &rng, uint32_t lane, uint32_t ops[3], uint32_t cap, uint32_t *errw) {
    P.n = 0;
    const unsigned long long tb0 = clock64(); (void)tb0;
    uint32_t preci = M.n;
    for (uint32_t l = 0; l < M.m; ++l) {
        const uint32_t r = perm[l], b = M.rp[r], len = (uint32_t)M.rp[r + 1u] - b;
        if (len == 0 || len > 64u) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_ROW); return; }
        if (P.n + 2u * len + 2u > cap) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_CAP); return; }
        const bool neg = (sgn[l] >> sbit) & 1u;
        int c = -1, v = 0;
        if (lane < len) { c = M.col[b + lane]; v = M.val[b + lane]; if (neg) v = -v; }
        uint32_t ai;
        if (!oriented) ai = t_rng(rng) % len;                                                          // :226-232
        else {                                                                                         // :179-216
            const unsigned long long mp = __builtin_amdgcn_ballot_w64(lane < len && (uint32_t)c == preci);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(lane < len && v == 1);
            ai = mp ? (uint32_t)__builtin_ctzll(mp) : len;
            if (ai == len || !((m1 >> ai) & 1ull)) {
                const uint32_t cnt = (uint32_t)__builtin_popcountll(m1);
                if (cnt) {
                    uint32_t pick = t_rng(rng) % cnt; unsigned long long mm = m1;
                    while (pick--) mm &= mm - 1ull;
                    ai = (uint32_t)__builtin_ctzll(mm);
                }
            }
            if (ai == len) ai = 0;
        }
        const int i = __builtin_amdgcn_readlane(c, (int)ai), av = __builtin_amdgcn_readlane(v, (int)ai);
        const uint32_t scale = (!transposed && av != 1) ? 1u : 0u;                                      // transposed: only if not +-1
        const uint32_t base = P.n, bar = base + scale + (len - 1u);
        if (lane < len && lane != ai) {
            const uint32_t rk = lane - (lane > ai ? 1u : 0u);
            const uint32_t p1 = base + scale + rk, p2 = bar + 1u + rk;
            if (transposed) {                                                                          // :424-427, :444-447
                P.src[p1] = (uint16_t)c; P.des[p1] = (int16_t)i; P.val[p1] = (int16_t)v; P.ope[p1] = (uint8_t)(av == -1 ? T_ADD : T_SUB);
                P.src[p2] = (uint16_t)c; P.des[p2] = (int16_t)i; P.val[p2] = (int16_t)v; P.ope[p2] = (uint8_t)(av == -1 ? T_SUB : T_ADD);
            } else {                                                                                   // :428-431, :448-451
                P.src[p1] = (uint16_t)i; P.des[p1] = (int16_t)c; P.val[p1] = (int16_t)v; P.ope[p1] = T_ADD;
                P.src[p2] = (uint16_t)i; P.des[p2] = (int16_t)c; P.val[p2] = (int16_t)v; P.ope[p2] = T_SUB;
            }
        }
        if (lane == ai) {
            if (scale) { P.src[base] = (uint16_t)i; P.des[base] = -1; P.val[base] = (int16_t)av; P.ope[base] = T_MUL; }
            P.src[bar] = (uint16_t)i; P.des[bar] = -1; P.val[bar] = (int16_t)av; P.ope[bar] = T_BAR;
            if (scale) { const uint32_t u = bar + len; P.src[u] = (uint16_t)i; P.des[u] = -1; P.val[u] = (int16_t)av; P.ope[u] = T_DIV; }
        }
        P.n = base + 2u * scale + 2u * (len - 1u) + 1u;
        if (len > 1u) preci = (uint32_t)i;
        TW_SYNC();
    }
    // no '*1' atoms exist for +-1 inputs (:481-482); fixpoint :488-494
    TP_ADD(1, tb0);
    bool simp;
    do {
        unsigned long long t1 = clock64();
        if (transposed) t_pushvariables(P, M.n, lane);
        TP_ADD(2, t1); t1 = clock64();
        simp = t_simplify(P, transposed, lane);
        TP_ADD(3, t1);
#ifdef PLO_TRIL_PROFILE
        if (threadIdx.x == 0) g_tprof[4] += 1;
#endif
    } while (simp);
    uint32_t a = 0, s = 0, mu = 0;                                                                    // :133-144
    for (uint32_t k = lane; k < P.n; k += 64u) {
        const uint32_t o = P.ope[k]; const int v = P.val[k];
        if (t_as(o)) { ++a; if (v != 1 && v != -1) ++s; }
        if (t_md(o)) ++s;
        if (o == T_BAR) ++mu;
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); s += __shfl_xor(s, off); mu += __shfl_xor(mu, off); }
    ops[0] = a; ops[1] = s; ops[2] = mu;
}

__global__ __launch_bounds__(256) void tril_kernel(TrilPlan P, TrilJob J)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tdyn[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint8_t *reg = tdyn + (size_t)wave * P.lds_per_wave;
    const uint32_t cap = P.cap, m = P.M[0].m;
    TrilProg G;
    G.src = (uint16_t *)reg; G.des = (int16_t *)(reg + 2u * cap); G.val = (int16_t *)(reg + 4u * cap); G.ope = reg + 6u * cap; G.n = 0;
    uint16_t *perm = (uint16_t *)(reg + 7u * cap);
    uint8_t *sgn = reg + 7u * cap + 2u * ((m + 1u) & ~1u);
    unsigned long long best = ~0ull;
    const uint64_t stride = (uint64_t)gridDim.x * nw;
    for (uint64_t cnd = (uint64_t)blockIdx.x * nw + wave; cnd < J.ncand; cnd += stride) {
        const uint64_t seed = J.seeds ? J.seeds[cnd] : J.seed0 + cnd;
        const unsigned long long tc0 = clock64(); (void)tc0;
        uint32_t rng = 1u + (uint32_t)(t_splitmix(seed) % 2147483646ull);
        const bool basec = seed == ~0ull;
        for (uint32_t k = lane; k < m; k += 64u) { perm[k] = (uint16_t)k; sgn[k] = 0; }
        TW_SYNC();
        if (!basec) {
            // the stream is sequential: one lane draws (Fisher-Yates :842-844, brand() pairs :872-885)
            if (lane == 0) {
                for (uint32_t i = m; i > 1u; --i) { const uint32_t j = t_rng(rng) % i; const uint16_t t = perm[i - 1u]; perm[i - 1u] = perm[j]; perm[j] = t; }
                for (uint32_t i = 0; i < m; ++i) { const uint32_t na = t_rng(rng) & 1u, nb = t_rng(rng) & 1u; sgn[i] = (uint8_t)(na | (nb << 1) | ((na ^ nb) << 2)); }
            }
            rng = t_uni(rng);
            TW_SYNC();
        }
        TP_ADD(0, tc0);
#ifdef PLO_TRIL_PROFILE
        if (threadIdx.x == 0) g_tprof[5] += 1;
#endif
        uint32_t tot[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t variant = 0; variant < 2u; ++variant) {
            if (basec && variant == 1u) { tot[3] = tot[0]; tot[4] = tot[1]; tot[5] = tot[2]; break; }
            for (uint32_t w = 0; w < 3u; ++w) {
                uint32_t o[3] = {0, 0, 0};
                t_linear(G, P.M[w], perm, sgn, w, w == 2u, variant == 0u, rng, lane, o, cap, J.err);
                tot[3u * variant] += o[0]; tot[3u * variant + 1u] += o[1]; tot[3u * variant + 2u] += o[2];
            }
            tot[3u * variant + 2u] /= 3u;                                                            // :801-803
        }
        if (lane == 0) {
            if (J.ops) for (int k = 0; k < 6; ++k) J.ops[6u * cnd + k] = tot[k];
            for (uint32_t variant = 0; variant < 2u; ++variant) {
                const unsigned long long key = ((unsigned long long)(tot[3u * variant] & 0xFFFFu) << 48) | ((unsigned long long)(tot[3u * variant + 1u] & 0xFFFFu) << 32)
                                             | ((unsigned long long)(cnd & 0x7FFFFFFFull) << 1) | variant;
                best = key < best ? key : best;
            }
        }
    }
    if (J.best && lane == 0 && best != ~0ull) atomicMin(J.best, best);
}

} // namespace plo
