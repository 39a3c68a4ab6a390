// ==========================================================================
// plo_tril.hip -- in-place trilinear program search (trilplacer) on gfx950.
//
// Replaces the body of the restart loop of SearchTriLinearAlgorithm
// (reference include/plinopt_inplace.inl:837-924): one candidate = a row permutation
// and coherent row negations of (A, B, T), then the oriented and the unoriented
// in-place program (LinearAlgorithm :400-502 on A, B and, transposed, on T) and
// their (ADD, SCA) counts (complexity :133-144).  Counts only: the host replays the
// winning seed to print the program.  `trilplacer -e`: the program of T is
// TransposedDoubleAlgorithm (:507-598) on the double expansion of T (t_double).
//
// One wavefront per candidate.  The atom list of the program being built lives in
// LDS, one 8-byte word per atom (src, des, val, ope); the control flow is wave
// uniform and the lanes share the scans:
//   simplify  (:243-311)  64 atoms at a time, every lane walks the atoms after its
//                         own one until it meets a mergeable atom or a dependency;
//                         the first lane (program order) that found a merge applies it;
//   pushvariables (:322-393)  per output variable, "next atom of the variable" and
//                         "first atom that stops the push" are ballot scans; a rotation
//                         is a parallel shift of the LDS arrays.
// Coefficients: +-1 (small signed values in the atoms) or rationals as residues modulo a
// 31-bit prime (tril_kernel<true>), `-e` included; an empty row or a row of more than 64
// entries is reported as unsupported and stays on the host.
// ==========================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

struct TrilMat { uint32_t m, n, nnz; const uint16_t *rp; const uint16_t *col; const int8_t *val; const uint32_t *valp; };   // valp: residues modulo TrilPlan::p (rational inputs)
struct TrilPlan { TrilMat M[3]; uint32_t cap; uint32_t lds_per_wave; uint32_t expanded; uint32_t p; };   // expanded: `trilplacer -e`; p != 0: rational coefficients as residues modulo the prime p
struct TrilJob {
    uint64_t seed0; const uint64_t *seeds; uint64_t ncand;
    uint32_t *ops;               // 6 per candidate: ADD,SCA,MUL oriented then unoriented (may be null)
    unsigned long long *best;    // packed minimum (may be null)
    uint32_t *err;
};
enum { TERR_CAP = 21, TERR_ROW = 22 };
enum { T_BAR = 0, T_ADD = 1, T_SUB = 2, T_MUL = 3, T_DIV = 4 };

#ifndef PLO_TRIL_LOCKSTEPS
#define PLO_TRIL_LOCKSTEPS 20u
#endif
#define TW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

#ifdef PLO_TRIL_PROFILE
__device__ unsigned long long g_tprof[8];   // lane 0 of wave 0 of every workgroup: cycles in perm, build, pushvariables, simplify; calls
#define TP_ADD(k_, t0_) do { if (threadIdx.x == 0) g_tprof[k_] += clock64() - (t0_); } while (0)
#else
#define TP_ADD(k_, t0_) do { } while (0)
#endif
// one atom = one 8-byte word: src | des<<16 | val<<32 | ope<<48 (des, val signed 16 bits)
// Rational inputs (round 3; Atom::_val is a Givaro::Rational, plinopt_inplace.inl:19): the value of an atom is its image modulo a
// 31-bit prime, and the word is src (14 bits) | des (14 bits, all ones = none) | ope (3 bits) in the low half, the residue in the
// high half.  What the program's counts depend on survives the image: an additive atom carries the SIGNED coefficient of its
// operation (cumulate :96-107 adds or subtracts and then only normalises the sign: ope and |val| are a representation), it is a
// no-op iff the coefficient is 0 and scalar iff it is not +-1 (complexity :133-144); a multiplicative atom carries its factor,
// a no-op iff 1 (:108-118 normalises |val| >= 1 the same way).  A collision modulo the prime would need a non-zero rational
// of the inputs' size to vanish modulo 2147483629; the tool replays and verifies the winner over Q in any case.
struct TrilProg { uint64_t *at; uint32_t n; };
template <bool RAT> __device__ __forceinline__ uint64_t ta_make(uint32_t src, int des, uint32_t val, uint32_t ope) {
    if constexpr (RAT) return (uint64_t)((src & 0x3FFFu) | (((uint32_t)des & 0x3FFFu) << 14) | (ope << 28)) | ((uint64_t)val << 32);
    else return (uint64_t)(src & 0xFFFFu) | ((uint64_t)((uint32_t)des & 0xFFFFu) << 16) | ((uint64_t)(val & 0xFFFFu) << 32) | ((uint64_t)ope << 48);
}
template <bool RAT> __device__ __forceinline__ int ta_src(uint64_t a) { if constexpr (RAT) return (int)((uint32_t)a & 0x3FFFu); else return (int)(a & 0xFFFFull); }
template <bool RAT> __device__ __forceinline__ int ta_des(uint64_t a) {
    if constexpr (RAT) { const uint32_t d = ((uint32_t)a >> 14) & 0x3FFFu; return d == 0x3FFFu ? -1 : (int)d; }
    else return (int)(int16_t)(uint16_t)(a >> 16);
}
// unit inputs: a small signed integer; rational inputs: the residue (as int bits)
template <bool RAT> __device__ __forceinline__ int ta_val(uint64_t a) { if constexpr (RAT) return (int)(uint32_t)(a >> 32); else return (int)(int16_t)(uint16_t)(a >> 32); }
template <bool RAT> __device__ __forceinline__ uint32_t ta_ope(uint64_t a) { if constexpr (RAT) return ((uint32_t)a >> 28) & 7u; else return (uint32_t)(a >> 48) & 0xFFu; }
__device__ __forceinline__ uint32_t t_mulmod(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)(((uint64_t)a * b) % p); }
__device__ __forceinline__ uint32_t t_invmod(uint32_t a, uint32_t p) { uint32_t r = 1u, bs = a; for (uint32_t e = p - 2u; e; e >>= 1) { if (e & 1u) r = t_mulmod(r, bs, p); bs = t_mulmod(bs, bs, p); } return r; }

__device__ __forceinline__ uint32_t t_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t t_rng(uint32_t &s) { s = (uint32_t)((950706376ull * (uint64_t)s) % 2147483647ull); return s; }
__device__ __forceinline__ uint64_t t_splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__device__ __forceinline__ bool t_as(uint32_t o) { return o == T_ADD || o == T_SUB; }
__device__ __forceinline__ bool t_md(uint32_t o) { return o == T_MUL || o == T_DIV; }

// [lo, hi) moves one slot down onto [lo-1, hi-1) (64 at a time, lowest chunk first: a chunk's reads are complete
// before its writes, and it reads the first element of the next chunk before that chunk is written)
__device__ __forceinline__ void t_shift_down(TrilProg &P, uint32_t lo, uint32_t hi, uint32_t lane) {
    for (uint32_t b = lo; b < hi; b += 64u) {
        const uint32_t d = b + lane; const bool on = d < hi;
        uint64_t a = 0;
        if (on) a = P.at[d];
        TW_SYNC();
        if (on) P.at[d - 1u] = a;
        TW_SYNC();
    }
}
__device__ __forceinline__ void t_erase(TrilProg &P, uint32_t k, uint32_t lane) { t_shift_down(P, k + 1u, P.n, lane); --P.n; }
// std::rotate(f, f+1, e): atom f goes to e-1, [f+1, e) moves one slot down
__device__ __forceinline__ void t_rotate(TrilProg &P, uint32_t f, uint32_t e, uint32_t lane) {
    const uint64_t fa = P.at[f];
    TW_SYNC();
    t_shift_down(P, f + 1u, e, lane);
    if (lane == 0) P.at[e - 1u] = fa;
    TW_SYNC();
}

// first k in [start, n) with pred(atom k), or n
template <class F> __device__ __forceinline__ uint32_t t_find(const TrilProg &P, uint32_t start, uint32_t n, uint32_t lane, F pred) {
    for (uint32_t b = start; b < n; b += 64u) {
        const uint32_t k = b + lane;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(k < n && pred(P.at[k < n ? k : start]));
        if (m) return b + (uint32_t)__builtin_ctzll(m);
    }
    return n;
}

// what stops the walk of atom `it` at atom `nx`: a merge (sameops + cumulate, :266-282) or a dependency (:284-307)
template <bool RAT> __device__ __forceinline__ bool t_merges(uint64_t it, uint64_t nx) {
    const uint32_t io = ta_ope<RAT>(it), no = ta_ope<RAT>(nx);
    const bool same = RAT ? (((uint32_t)it ^ (uint32_t)nx) & 0x0FFFFFFFu) == 0u : (uint32_t)it == (uint32_t)nx;
    return same && ((t_as(io) && t_as(no)) || (t_md(io) && t_md(no)));       // same src and des, compatible operations
}
template <bool RAT> __device__ __forceinline__ bool t_breaks(uint64_t it, uint64_t nx, bool transposed) {
    const int is = ta_src<RAT>(it), id = ta_des<RAT>(it), ns = ta_src<RAT>(nx), nd = ta_des<RAT>(nx); const uint32_t io = ta_ope<RAT>(it), no = ta_ope<RAT>(nx);
    bool brk = (is == ns) && (no == T_BAR || (t_as(io) && t_md(no)) || (t_md(io) && t_as(no)));
    brk |= transposed ? (id == ns) : (id == ns && no != T_BAR);
    brk |= (is == nd);
    return brk;
}

// :243-311.  true = one merge applied.
template <bool RAT> __device__ bool t_simplify(TrilProg &P, bool transposed, uint32_t lane, uint32_t prime) {
    for (uint32_t base = 0; base < P.n; base += 64u) {
        const uint32_t p = base + lane, n = P.n;
        uint64_t it = 0;
        bool act = p < n;
        if (act) { it = P.at[p]; act = ta_ope<RAT>(it) != T_BAR; }
        uint32_t k = p + 1u, q = 0; int res = act ? 0 : 2;       // 0 walking, 1 merge found, 2 stopped
        // most walks end within a few atoms: a few lock-step steps, one atom per lane ...
        for (uint32_t step = 0; step < PLO_TRIL_LOCKSTEPS && __builtin_amdgcn_ballot_w64(res == 0); ++step) {
            if (res == 0) {
                if (k >= n) res = 2;
                else {
                    const uint64_t nx = P.at[k];
                    if (t_merges<RAT>(it, nx)) { res = 1; q = k; }
                    else if (t_breaks<RAT>(it, nx, transposed)) res = 2;
                    else ++k;
                }
            }
        }
        // ... and each long walk that is still open before the first merge found so far is finished by the whole wave
        {
            const unsigned long long ok1 = __builtin_amdgcn_ballot_w64(res == 1);
            unsigned long long und = __builtin_amdgcn_ballot_w64(res == 0);
            if (ok1) und &= (1ull << __builtin_ctzll(ok1)) - 1ull;
            while (und) {
                const int L = __builtin_ctzll(und); und &= und - 1ull;
                const uint64_t bit = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(it >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)it, L);
                const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)k, L);
                const uint32_t e = t_find(P, bk, n, lane, [&](uint64_t nx) { return t_merges<RAT>(bit, nx) || t_breaks<RAT>(bit, nx, transposed); });
                const bool mg = e < n && t_merges<RAT>(bit, P.at[e < n ? e : 0u]);
                if ((int)lane == L) { if (mg) { res = 1; q = e; } else res = 2; }
                if (mg) break;
            }
        }
        const unsigned long long ok = __builtin_amdgcn_ballot_w64(res == 1);
        if (ok) {
            const int L = __builtin_ctzll(ok);
            const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)p, L), qq = (uint32_t)__builtin_amdgcn_readlane((int)q, L);
            const uint64_t a1 = P.at[pp], a2 = P.at[qq];
            const uint32_t o1 = ta_ope<RAT>(a1), o2 = ta_ope<RAT>(a2); const int v1 = ta_val<RAT>(a1), v2 = ta_val<RAT>(a2);
            uint32_t o = o1; int v; bool noop;
            if constexpr (RAT) {
                const uint32_t u1 = (uint32_t)v1, u2 = (uint32_t)v2; uint32_t u;
                if (t_as(o1)) { u = (o1 == o2) ? (uint32_t)(((uint64_t)u1 + u2) % prime) : (uint32_t)(((uint64_t)u1 + prime - u2) % prime); noop = u == 0u; }   // :96-107: the signed coefficient of o1 (the sign swap of :103-106 is a representation)
                else { u = (o1 == o2) ? t_mulmod(u1, u2, prime) : t_mulmod(u1, t_invmod(u2, prime), prime); noop = u == 1u; }                                  // :108-118
                v = (int)u;
            } else if (t_as(o1)) {                            // :96-107
                v = (o1 == o2) ? v1 + v2 : v1 - v2;
                if (v < 0) { o = (o1 == T_ADD) ? T_SUB : T_ADD; v = -v; }
                noop = v == 0;
            } else {                                          // :108-118, values are +-1
                v = v1 * v2; noop = v == 1;
            }
            TW_SYNC();
            t_erase(P, qq, lane);
            if (noop) t_erase(P, pp, lane);
            else if (lane == 0) P.at[pp] = ta_make<RAT>((uint32_t)ta_src<RAT>(a1), ta_des<RAT>(a1), (uint32_t)v, o);
            TW_SYNC();
            return true;
        }
    }
    return false;
}

// :322-393, one pass per variable instead of one (find, find, rotate) trip per couple.
// Within the pass of variable i the reference looks for the next atom f of the variable from e + 1, the atom e that stops the
// push of f depends only on atoms after f, and std::rotate(f, f+1, e) touches [f, e): the couples (f, e) of a pass can all be
// found on the list as it stands and their rotations, which are disjoint, applied as ONE permutation.
//   1. the atoms of the variable, in order, one per lane (ballots + prefix popcount);
//   2. every such lane walks from its atom to its stopper (lock step, then the whole wave finishes the long walks);
//   3. the chain f_1, e_1, f_2 = first atom of the variable behind e_1, ... is followed on the lanes' registers;
//   4. the selected intervals become two bit masks (starts f+1, ends e) in LDS: position x moves down one slot iff
//      #starts <= x exceeds #ends <= x; the atoms f are written to e - 1 at the end.
// bm: 4 * ceil(cap/64) words of LDS (start and end masks).  Same result as t_pushvariables_ref below (bit-exact on the 49 cases
// of tests/test_gpu_tril.py) -- and NOT faster: measured on 4x4x4_49_156 {L,R,P} (MI355X, -DPLO_TRIL_PROFILE) 11.35 M cycles of
// pushvariables per candidate against 9.42 M for the literal form (5.13e5 against 5.70e5 candidates/s).  A pass costs ~1,300
// wave instructions whichever way it is organised (gather 270, lock-step walks 500, chain 340, permutation 225) and the kernel
// is bound by instruction issue at 8 waves per SIMD: the couples were never the cost, the 71 fixpoint trips x 16 passes are.
// Built with -DPLO_TRIL_PASSPUSH; the default is the literal form.
template <bool RAT> __device__ void t_pushvariables(TrilProg &P, uint32_t numout, uint32_t lane, uint32_t *bm) {
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t i = 0; i < numout; ++i) {
        uint32_t pos = 0;
        for (;;) {                                                              // (one round per 64 atoms of the variable: almost always one)
            const uint32_t n = P.n, nch = (n + 63u) >> 6;
            // 1. atoms of the variable at positions >= pos: lane j gets the j-th
            uint32_t c = n, cnt = 0; uint64_t fa = 0;
            for (uint32_t b = pos & ~63u; b < n && cnt < 64u; b += 64u) {
                const uint32_t k = b + lane;
                uint64_t a = 0; bool is = false;
                if (k < n && k >= pos) { a = P.at[k]; is = ta_ope<RAT>(a) != T_BAR && ta_src<RAT>(a) == (int)i; }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(is);
                if (!m) continue;
                // lane (cnt + rank) takes the atom of the lane with that rank in m
                const uint32_t want = lane - cnt;                              // rank this lane would take (if < popcount)
                const uint32_t np = (uint32_t)__builtin_popcountll(m);
                // source lane of rank `want`: select the want-th set bit of m
                uint32_t srcl = 0; { unsigned long long mm = m; uint32_t w = want < np ? want : 0u; while (w--) mm &= mm - 1ull; srcl = (uint32_t)__builtin_ctzll(mm); }
                const uint32_t ak = (uint32_t)__shfl((int)k, (int)srcl);
                const uint32_t alo = (uint32_t)__shfl((int)(uint32_t)a, (int)srcl), ahi = (uint32_t)__shfl((int)(uint32_t)(a >> 32), (int)srcl);
                if (lane >= cnt && want < np) { c = ak; fa = ((uint64_t)ahi << 32) | alo; }
                cnt += np;
            }
            if (cnt == 0) break;
            const uint32_t ncand = cnt < 64u ? cnt : 64u;
            const bool have = lane < ncand;
            // 2. stoppers
            const uint32_t fo = ta_ope<RAT>(fa); const int fd = ta_des<RAT>(fa); const bool fas = t_as(fo);
            auto stops = [&](uint64_t a) -> bool {
                const int sa = ta_src<RAT>(a);
                return fas ? (fd == sa || (sa == (int)i && (fd == ta_des<RAT>(a) || t_md(ta_ope<RAT>(a))))) : (ta_des<RAT>(a) == (int)i || sa == (int)i);
            };
            uint32_t k = c + 1u, e = n; int res = have ? 0 : 2;               // 0 walking, 2 done (e = stopper or n)
            for (uint32_t step = 0; step < PLO_TRIL_LOCKSTEPS && __builtin_amdgcn_ballot_w64(res == 0); ++step) {
                if (res == 0) {
                    if (k >= n) res = 2;
                    else if (stops(P.at[k])) { e = k; res = 2; }
                    else ++k;
                }
            }
            {
                unsigned long long und = __builtin_amdgcn_ballot_w64(res == 0);
                while (und) {
                    const int L = __builtin_ctzll(und); und &= und - 1ull;
                    const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)k, L);
                    const int bfd = __builtin_amdgcn_readlane(fd, L); const bool bfas = __builtin_amdgcn_readlane((int)fas, L) != 0;
                    const uint32_t ee = t_find(P, bk, n, lane, [&](uint64_t a) { const int sa = ta_src<RAT>(a);
                        return bfas ? (bfd == sa || (sa == (int)i && (bfd == ta_des<RAT>(a) || t_md(ta_ope<RAT>(a))))) : (ta_des<RAT>(a) == (int)i || sa == (int)i); });
                    if ((int)lane == L) { e = ee; res = 2; }
                }
            }
            bool rot = false;
            if (have && e < n) { const uint64_t ea = P.at[e]; rot = fas ? (!(fd == ta_src<RAT>(ea)) && fd == ta_des<RAT>(ea)) : (!(ta_des<RAT>(ea) == (int)i) && t_md(ta_ope<RAT>(ea))); }
            // 3. the chain of couples
            unsigned long long sel = 0; uint32_t cur = 0, nextpos = n; bool passdone = false;
            for (;;) {
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)cur), ee = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)cur);
                const bool rr = __builtin_amdgcn_readlane((int)rot, (int)cur) != 0;
                if (ee >= n) { if (f + 1u != n) sel |= 1ull << cur; passdone = true; break; }     // can be moved to the end (:381-391)
                if (rr && f + 1u != ee) sel |= 1ull << cur;
                const unsigned long long nx = __builtin_amdgcn_ballot_w64(have && c > ee);
                if (!nx) { nextpos = ee + 1u; passdone = cnt <= 64u; break; }                      // no further atom of the variable in this round
                cur = (uint32_t)__builtin_ctzll(nx);
            }
            // 4. one permutation for the selected couples
            if (sel) {
                const bool mine = (sel >> lane) & 1ull;
                for (uint32_t w = lane; w < 4u * nch; w += 64u) bm[w] = 0u;
                TW_SYNC();
                if (mine) {
                    const uint32_t st = c + 1u;
                    atomicOr(&bm[st >> 5], 1u << (st & 31u));
                    if (e < n) atomicOr(&bm[2u * nch + (e >> 5)], 1u << (e & 31u));
                }
                TW_SYNC();
                uint32_t carry = 0;
                for (uint32_t ch = 0; ch < nch; ++ch) {
                    const uint32_t x = (ch << 6) + lane;
                    const unsigned long long S = ((unsigned long long)bm[2u * ch + 1u] << 32) | bm[2u * ch], E = ((unsigned long long)bm[2u * nch + 2u * ch + 1u] << 32) | bm[2u * nch + 2u * ch];
                    const unsigned long long upto = below | (1ull << lane);
                    const bool inside = (int)(carry + (uint32_t)__builtin_popcountll(S & upto)) - (int)__builtin_popcountll(E & upto) > 0;
                    uint64_t a = 0;
                    if (x < n && inside) a = P.at[x];
                    TW_SYNC();
                    if (x < n && inside) P.at[x - 1u] = a;
                    TW_SYNC();
                    carry += (uint32_t)__builtin_popcountll(S) - (uint32_t)__builtin_popcountll(E);
                }
                if (mine) P.at[(e < n ? e : n) - 1u] = fa;
                TW_SYNC();
            }
            if (passdone) break;
            pos = nextpos;
        }
    }
}

// the literal form: one (find, find, rotate) trip per couple
template <bool RAT> __device__ void t_pushvariables_ref(TrilProg &P, uint32_t numout, uint32_t lane) {
    for (uint32_t i = 0; i < numout; ++i) {
        uint32_t pos = 0;
        for (;;) {
            const uint32_t n = P.n;
            const uint32_t f = t_find(P, pos, n, lane, [&](uint64_t a) { return ta_ope<RAT>(a) != T_BAR && ta_src<RAT>(a) == (int)i; });
            if (f >= n) break;
            const uint64_t fa = P.at[f];
            const uint32_t fo = ta_ope<RAT>(fa); const int fd = ta_des<RAT>(fa);
            uint32_t e;
            if (t_as(fo)) e = t_find(P, f + 1u, n, lane, [&](uint64_t a) { const int s = ta_src<RAT>(a); return fd == s || (s == (int)i && (fd == ta_des<RAT>(a) || t_md(ta_ope<RAT>(a)))); });
            else e = t_find(P, f + 1u, n, lane, [&](uint64_t a) { return ta_des<RAT>(a) == (int)i || ta_src<RAT>(a) == (int)i; });
            if (e >= n) { if (f + 1u != n) t_rotate(P, f, n, lane); break; }      // can be moved to the end (:381-391)
            const uint64_t ea = P.at[e];
            bool rot;
            if (t_as(fo)) rot = !(fd == ta_src<RAT>(ea)) && fd == ta_des<RAT>(ea);
            else rot = !(ta_des<RAT>(ea) == (int)i) && t_md(ta_ope<RAT>(ea));
            if (rot && f + 1u != e) t_rotate(P, f, e, lane);
            pos = e + 1u;
        }
    }
}

// :400-502 for matrices without empty rows; perm/sign describe the candidate's rows.  RAT = false: entries +-1 (small signed values,
// no scaling atom in the transposed program); RAT = true: entries are residues of rationals modulo `prime`.
template <bool RAT> __device__ void t_linear(TrilProg &P, const TrilMat &M, const uint16_t *perm, const uint8_t *sgn, uint32_t sbit, bool transposed,
                         bool oriented, uint32_t &rng, uint32_t lane, uint32_t ops[3], uint32_t cap, uint32_t *errw, uint32_t *bm, uint32_t prime) {
    P.n = 0;
    const unsigned long long tb0 = clock64(); (void)tb0;
    uint32_t preci = M.n;
    const int ONE = 1, MONE = RAT ? (int)(prime - 1u) : -1;
    for (uint32_t l = 0; l < M.m; ++l) {
        const uint32_t r = perm[l], b = M.rp[r], len = (uint32_t)M.rp[r + 1u] - b;
        if (len == 0 || len > 64u) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_ROW); return; }
        if (P.n + 2u * len + 2u > cap) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_CAP); return; }
        const bool neg = (sgn[l] >> sbit) & 1u;
        int c = -1, v = 0;
        if (lane < len) {
            c = M.col[b + lane];
            if constexpr (RAT) { const uint32_t x = M.valp[b + lane]; v = (int)(neg ? (x ? prime - x : 0u) : x); }
            else { v = M.val[b + lane]; if (neg) v = -v; }
        }
        uint32_t ai;
        if (!oriented) ai = t_rng(rng) % len;                                                          // :226-232
        else {                                                                                         // :179-216
            const unsigned long long mp = __builtin_amdgcn_ballot_w64(lane < len && (uint32_t)c == preci);
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(lane < len && v == ONE);
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
        // scaling of the chosen variable (:416-424, :467-475): direct program: whenever the pivot is not 1; transposed: unless it is +-1
        const uint32_t scale = transposed ? ((av != ONE && av != MONE) ? 1u : 0u) : (av != ONE ? 1u : 0u);
        const uint32_t base = P.n, bar = base + scale + (len - 1u);
        if (lane < len && lane != ai) {
            const uint32_t rk = lane - (lane > ai ? 1u : 0u);
            const uint32_t p1 = base + scale + rk, p2 = bar + 1u + rk;
            if (transposed) {                                                                          // :428-431, :448-451 (MONEOP: swapped when the pivot is -1)
                P.at[p1] = ta_make<RAT>((uint32_t)c, i, (uint32_t)v, av == MONE ? T_ADD : T_SUB);
                P.at[p2] = ta_make<RAT>((uint32_t)c, i, (uint32_t)v, av == MONE ? T_SUB : T_ADD);
            } else {                                                                                   // :432-435, :452-455
                P.at[p1] = ta_make<RAT>((uint32_t)i, c, (uint32_t)v, T_ADD);
                P.at[p2] = ta_make<RAT>((uint32_t)i, c, (uint32_t)v, T_SUB);
            }
        }
        if (lane == ai) {
            if (scale) P.at[base] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)av, transposed ? T_DIV : T_MUL);
            P.at[bar] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)av, T_BAR);
            if (scale) P.at[bar + len] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)av, transposed ? T_MUL : T_DIV);
        }
        P.n = base + 2u * scale + 2u * (len - 1u) + 1u;
        if (len > 1u) preci = (uint32_t)i;
        TW_SYNC();
    }
    // no '*1' atoms exist (a scaling atom is only made for a pivot other than 1, :481-482); fixpoint :488-494
    TP_ADD(1, tb0);
    bool simp;
    do {
        unsigned long long t1 = clock64();
#ifdef PLO_TRIL_PASSPUSH
        if (transposed) t_pushvariables<RAT>(P, M.n, lane, bm);
#else
        if (transposed) t_pushvariables_ref<RAT>(P, M.n, lane);
#endif
        TP_ADD(2, t1); t1 = clock64();
        simp = t_simplify<RAT>(P, transposed, lane, prime);
        TP_ADD(3, t1);
#ifdef PLO_TRIL_PROFILE
        if (threadIdx.x == 0) g_tprof[4] += 1;
#endif
    } while (simp);
    uint32_t a = 0, s = 0, mu = 0;                                                                    // :133-144
    for (uint32_t k = lane; k < P.n; k += 64u) {
        const uint64_t at = P.at[k]; const uint32_t o = ta_ope<RAT>(at); const int v = ta_val<RAT>(at);
        if (t_as(o)) { ++a; if (v != ONE && v != MONE) ++s; }
        if (t_md(o)) ++s;
        if (o == T_BAR) ++mu;
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); s += __shfl_xor(s, off); mu += __shfl_xor(mu, off); }
    ops[0] = a; ops[1] = s; ops[2] = mu;
}

// `trilplacer -e`: TransposedDoubleAlgorithm (:507-598) on DoubleExpand(T) (:676-716).  Row l of T stands for the pair of rows
// (2l, 2l+1) of the expanded matrix -- the block <<a|c>,<0|a>> on the variables i = first column and i+1 -- and gives, in the
// reference's order (:532-571): the scaling by y = 1/a (`*y` on i+1, the atom of z = -y c y when the row holds column i+1, `*y` on
// i; no scaling atom when a = +-1), two atoms per further entry (one for the entry at i+1), the two barriers of one double-size
// AXPY, the same atoms with the opposite sign, and the un-scaling (`*a`, the atom of c, `*a`).  One trip per pair of expanded
// rows: see oracle/plo_tril_oracle.c.  No random draw: the first entry is the pivot.  RAT = false: entries +-1 (y = a, z = -c);
// RAT = true (round 4): residues of rationals modulo `prime`, as in t_linear.
template <bool RAT> __device__ void t_double(TrilProg &P, const TrilMat &M, const uint16_t *perm, const uint8_t *sgn, uint32_t sbit, uint32_t lane,
                         uint32_t ops[3], uint32_t cap, uint32_t *errw, uint32_t *bm, uint32_t prime) {
    P.n = 0;
    const int ONE = 1, MONE = RAT ? (int)(prime - 1u) : -1;
    for (uint32_t l = 0; l < M.m; ++l) {
        const uint32_t r = perm[l], b = M.rp[r], len = (uint32_t)M.rp[r + 1u] - b;
        if (len == 0 || len > 64u) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_ROW); return; }
        if (P.n + 4u * len + 4u > cap) { if (lane == 0) atomicMax(errw, (uint32_t)TERR_CAP); return; }
        const bool neg = (sgn[l] >> sbit) & 1u;
        int c = -1, v = 0;
        if (lane < len) {
            c = M.col[b + lane];
            if constexpr (RAT) { const uint32_t x = M.valp[b + lane]; v = (int)(neg ? (x ? prime - x : 0u) : x); }
            else { v = M.val[b + lane]; if (neg) v = -v; }
        }
        const int i = __builtin_amdgcn_readlane(c, 0), a = __builtin_amdgcn_readlane(v, 0), ci = i + 1;
        const int c1 = __builtin_amdgcn_readlane(c, 1), v1 = __builtin_amdgcn_readlane(v, 1);
        const bool has_c = len > 1u && c1 == ci;                                                      // :524-529
        const bool amone = a == MONE;
        const uint32_t sc = (a != ONE && !amone) ? 1u : 0u;                                            // notAbsOne(y) <=> notAbsOne(a)
        int y = a, z = -v1;                                                                            // +-1: y = a, z = -a c a = -c
        if constexpr (RAT) {
            y = (int)t_invmod((uint32_t)a, prime);
            const uint32_t yc = t_mulmod(t_mulmod((uint32_t)y, (uint32_t)v1, prime), (uint32_t)y, prime);
            z = (int)(yc ? prime - yc : 0u);                                                          // z = - a^-1 c a^-1 (:527-529)
        }
        const uint32_t base = P.n, e0 = base + 2u * sc + (has_c ? 1u : 0u), bar = base + 2u * sc + 2u * (len - 1u), base2 = bar + 2u;
        const uint32_t end2 = base2 + 2u * (len - 1u) - (has_c ? 1u : 0u);
        const uint32_t o1 = amone ? T_ADD : T_SUB;             // MONEOP('-', y): y is -1 iff a is
        const uint32_t o2 = amone ? T_SUB : T_ADD;             // MONEOP('+', a) and MONEOP('+', y)
        if (lane >= 1u && lane < len) {
            if (has_c && lane == 1u) {
                P.at[e0] = ta_make<RAT>((uint32_t)(c + 1), ci, (uint32_t)v, o1);                             // :541-542 (the entry at i+1 only moves to i+2)
                P.at[base2] = ta_make<RAT>((uint32_t)(c + 1), ci, (uint32_t)v, o2);                          // :556-557
            } else {
                const uint32_t off = has_c ? 2u * lane - 3u : 2u * lane - 2u;
                P.at[e0 + off] = ta_make<RAT>((uint32_t)c, i, (uint32_t)v, o1);                              // :537-540
                P.at[e0 + off + 1u] = ta_make<RAT>((uint32_t)(c + 1), ci, (uint32_t)v, o1);                  // :541-542
                P.at[base2 + off] = ta_make<RAT>((uint32_t)c, i, (uint32_t)v, o2);                           // :552-555
                P.at[base2 + off + 1u] = ta_make<RAT>((uint32_t)(c + 1), ci, (uint32_t)v, o2);               // :556-557
            }
        }
        if (lane == 0) {
            if (sc) { P.at[base] = ta_make<RAT>((uint32_t)ci, -1, (uint32_t)y, T_MUL); P.at[base + 1u + (has_c ? 1u : 0u)] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)y, T_MUL); }       // :532, :535
            if (has_c) {
                P.at[base + sc] = ta_make<RAT>((uint32_t)ci, i, (uint32_t)z, o2);                            // :533-534
                P.at[end2 + sc] = ta_make<RAT>((uint32_t)ci, i, (uint32_t)v1, o2);                           // :563-564
            }
            P.at[bar] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)a, T_BAR);                                    // :546-548
            P.at[bar + 1u] = ta_make<RAT>((uint32_t)ci, -1, (uint32_t)a, T_BAR);
            if (sc) { P.at[end2] = ta_make<RAT>((uint32_t)ci, -1, (uint32_t)a, T_MUL); P.at[end2 + 1u + (has_c ? 1u : 0u)] = ta_make<RAT>((uint32_t)i, -1, (uint32_t)a, T_MUL); }         // :561, :565
        }
        P.n = base + 4u * sc + 4u * (len - 1u) + 2u;
        TW_SYNC();
    }
    // (no `*1` atom exists: a scaling atom is only made when a is not +-1, :586-587)
    bool simp;
#ifdef PLO_TRIL_PASSPUSH
    do { t_pushvariables<RAT>(P, M.n + 1u, lane, bm); simp = t_simplify<RAT>(P, true, lane, prime); } while (simp);
#else
    do { t_pushvariables_ref<RAT>(P, M.n + 1u, lane); simp = t_simplify<RAT>(P, true, lane, prime); } while (simp);
#endif
    uint32_t ad = 0, sca = 0, mu = 0;                                                                  // :133-144
    for (uint32_t k = lane; k < P.n; k += 64u) {
        const uint64_t at = P.at[k]; const uint32_t o = ta_ope<RAT>(at); const int v = ta_val<RAT>(at);
        if (t_as(o)) { ++ad; if (v != ONE && v != MONE) ++sca; }
        if (t_md(o)) ++sca;
        if (o == T_BAR) ++mu;
    }
    for (int off = 32; off > 0; off >>= 1) { ad += __shfl_xor(ad, off); sca += __shfl_xor(sca, off); mu += __shfl_xor(mu, off); }
    ops[0] = ad; ops[1] = sca; ops[2] = mu >> 1;                                                       // :799: a double-size AXPY holds two barriers
}

template <bool RAT> __global__ __launch_bounds__(256) void tril_kernel(TrilPlan P, TrilJob J)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tdyn[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint8_t *reg = tdyn + (size_t)wave * P.lds_per_wave;
    const uint32_t cap = P.cap, m = P.M[0].m;
    TrilProg G;
    G.at = (uint64_t *)reg; G.n = 0;
    uint16_t *perm = (uint16_t *)(reg + 8u * cap);
    uint8_t *sgn = reg + 8u * cap + 2u * ((m + 1u) & ~1u);
    uint32_t *bm = (uint32_t *)(reg + ((8u * cap + 2u * ((m + 1u) & ~1u) + m + 15u) & ~15u));   // start / end masks of a pushvariables pass: 4 words per 64 atoms
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
                bool done_ = false;
                if (w == 2u && P.expanded) { t_double<RAT>(G, P.M[2], perm, sgn, 2u, lane, o, cap, J.err, bm, P.p); done_ = true; }
                if (!done_) t_linear<RAT>(G, P.M[w], perm, sgn, w, w == 2u, variant == 0u, rng, lane, o, cap, J.err, bm, P.p);
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
