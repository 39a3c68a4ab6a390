// ===========================================================================
// plo_cse_wave.hip -- gfx950 kernel: one wavefront (64 lanes) evaluates one
// candidate of PLinOpt's randomized greedy pairwise-CSE search, counts only.
//
// Restates, per candidate, Optimizer() = { while(OneSub) ; ProgramGen }
// (reference include/plinopt_optimize.inl:616-631) over Z_p, p < 2^31:
//   OneSub      :209-314  pair table, max-frequency scan, random tie pick
//   RemOneCSE   :60-194   orientation by +-1 counts, row rewrite, table patch,
//                         multiplier reuse (`multiples`)
//   ProgramGen  :513-611  FactorOutColumns :318-371, FactorOutRows :375-420,
//                         Triangle :427-507, output rows :547-604
// The restart loop CSEOptimiser :1204-1238 is the grid: every wave walks its
// share of the seed range and keeps a packed (cost,seed) minimum; one 64-bit
// atomicMin per workgroup replaces the `#pragma omp critical` block :1214-1237.
//
// Data layout (per wave, all in LDS; the matrix image is fetched from an
// L2-resident template at the start of every candidate):
//   tab[cap]      u64  open-addressing pair table: key<<13 | count, key =
//                      col_a<<(bb+rb) | col_b<<rb | ratio (<= 51 bits), so integer order of
//                      keys == std::map order of the reference's (size_t,size_t,Element)
//   masks[NC*2mw] u64  per column c: cmask (bit i: row i holds column c) then umask
//                      (bit i: that entry is +-1); mw = ceil(m/64) words each
//   val/inv[nnz]  u32  row entries (value, its modular inverse), rows packed at
//   col[nnz]      u16  fixed offsets rs[i]; rows only shrink (2 entries -> 1)
//   len[m]        u16
//   ties[cap]     u16  slots of maximal frequency (scratch), mult[] multiplier list
// No MFMA: there is no dense contraction here; the work is LDS integer traffic.
// ===========================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

// waves per SIMD the wave kernels are compiled for (register budget 512 / waves): the kernels are LDS-latency bound, more
// resident waves hide more of it
#ifndef PLO_WAVE_OCC_UNIT
#define PLO_WAVE_OCC_UNIT 7
#endif
#ifndef PLO_WAVE_OCC_GEN
#define PLO_WAVE_OCC_GEN 4
#endif

#ifdef PLO_WAVE_PROFILE
__device__ unsigned long long g_wprof[12];   // lane 0 of every wave: cycles in max scan, tie list, tie pick, sweep 1, sweep 2, step tail, image load, ProgramGen; steps; candidates
#define WP_T(k_) do { const unsigned long long t__ = clock64(); wp_acc[k_] += t__ - wp_t; wp_t = t__; } while (0)
#else
#define WP_T(k_) do { } while (0)
#endif

struct WavePlan {
    uint32_t m, n, nnz, p, NC, cap, hbits, lpr_log2, mw, unit, multcap, maxlen;
    uint32_t rb, bb;        // bits of a ratio (p-1) and of a column index (NC-1): 2*bb+rb <= 51
    uint32_t off_tab, off_cmask, off_umask, off_val, off_inv, off_col, off_len;   // template part
    uint32_t tmpl_bytes;                                                          // multiple of 8
    uint32_t off_aff, off_ties, off_mult, region_bytes;                           // scratch part
    uint32_t rs_bytes;                                                            // shared rs[] image, multiple of 8
    uint64_t mu;                                                                  // floor(2^64/p)
    const uint64_t *tmpl;   // tmpl_bytes of template followed by rs_bytes of rs[m+1] (u16)
};

struct WaveJob {
    uint64_t seed0;            // candidate c has seed seed0 + c (seeds==nullptr) or seeds[c]
    const uint64_t *seeds;
    uint64_t ncand;
    uint32_t *adds, *muls;     // per-candidate outputs (may be null)
    unsigned long long *best;  // packed (cost,seed offset) minimum (may be null)
    uint32_t cost_mode;
    uint32_t *err;             // device error word
    uint32_t enumerate;        // 1: candidate c is the schedule of index seed0 + c in RecSub's tree (see PickState)
    unsigned long long *prodmax;   // enumerate: maximum over candidates of the product of their radices (may be null)
    unsigned long long *prods;     // enumerate: that product per candidate (may be null)
};

enum { ERR_TABLE = 1, ERR_MULT = 2, ERR_STEPS = 3, ERR_PGEN = 4 };

#define PLO_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = v > w ? v : w; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
        uint64_t w = ((uint64_t)hi << 32) | lo; v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v);
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t srclane) { return (uint32_t)__shfl((int)v, (int)srclane); }
// the same from a wave-uniform lane index: two v_readlane
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, uint32_t srclane) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)srclane) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)srclane);
}
__device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t srclane) {
    return ((uint64_t)bcast((uint32_t)(v >> 32), srclane) << 32) | bcast((uint32_t)v, srclane);
}

// position of the n-th (0-based) set bit of m; n < popcount(m)
__device__ __forceinline__ uint32_t nth_set_bit(uint64_t m, uint32_t n) {
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    uint32_t c = (uint32_t)__popc(lo), base = 0, wv = lo;
    if (n >= c) { n -= c; wv = hi; base = 32u; }
    c = (uint32_t)__popc(wv & 0xFFFFu); if (n >= c) { n -= c; wv >>= 16; base += 16u; }
    c = (uint32_t)__popc(wv & 0xFFu);   if (n >= c) { n -= c; wv >>= 8;  base += 8u; }
    c = (uint32_t)__popc(wv & 0xFu);    if (n >= c) { n -= c; wv >>= 4;  base += 4u; }
    c = (uint32_t)__popc(wv & 0x3u);    if (n >= c) { n -= c; wv >>= 2;  base += 2u; }
    c = wv & 1u;                        if (n >= c) base += 1u;
    return base;
}

template <bool UNIT>
__device__ __forceinline__ uint32_t fmul(uint32_t a, uint32_t b, uint32_t p, uint64_t mu) {
    if (UNIT) return a == b ? 1u : p - 1u;   // operands are +-1 only
    uint64_t x = (uint64_t)a * b;
    uint64_t q = __umul64hi(x, mu);
    uint64_t r = x - q * p;
    while (r >= p) r -= p;
    return (uint32_t)r;
}
__device__ __forceinline__ uint32_t fabsp(uint32_t e, uint32_t p) { uint32_t a = e ? p - e : 0u; return a < e ? a : e; }
__device__ __forceinline__ bool absone(uint32_t e, uint32_t p) { return e == 1u || e == p - 1u; }

// GivRandom LCG on the Mersenne prime 2^31-1
__device__ __forceinline__ uint32_t rng_next(uint32_t &s) {
    uint64_t x = 950706376ull * (uint64_t)s;
    x = (x & 0x7FFFFFFFull) + (x >> 31);
    x = (x & 0x7FFFFFFFull) + (x >> 31);
    if (x >= 0x7FFFFFFFull) x -= 0x7FFFFFFFull;
    s = (uint32_t)x; return s;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// value bits of a table slot: a frequency is a number of rows (<= 1984) and ProgramGen's multiset counts are too: 12 bits and one flag;
// the key has 51 bits, so that any odd prime below 2^31 leaves 10 bits per column (round 2: 20 value bits, 44-bit keys)
#define PLO_VB 13u
#define PLO_VMASK 0x1FFFull
#define PLO_EMPTY 0xFFFFFFFFFFFFE000ull             // key all ones, value 0
__device__ __forceinline__ uint32_t tab_hash(uint64_t key, uint32_t hbits) {
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu);
    return (x * 0x9E3779B1u) >> (32u - hbits);
}

// relaxed atomic read of a table slot: a `volatile` read of an LDS pointer is not address-space inferred and becomes a
// flat_load sc0 sc1 (measured in the ISA); this one compiles to ds_read_b64
__device__ __forceinline__ uint64_t lds_ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// count[key] -= 1 ; the key must be live
__device__ __forceinline__ bool tab_dec(uint64_t *tab, uint64_t key, uint32_t cap, uint32_t hbits) {
    uint32_t s = tab_hash(key, hbits);
    for (uint32_t pr = 0; pr < cap; ++pr) {
        uint64_t v = lds_ld64(&tab[s]);
        if ((v >> PLO_VB) == key) { atomicAdd((unsigned long long *)&tab[s], ~0ull); return true; }
        s = (s + 1u) & (cap - 1u);
    }
    return false;
}
// the same for two live keys at once: both probe sequences in flight together (one LDS round trip per round for the pair)
__device__ __forceinline__ bool tab_dec2(uint64_t *tab, uint64_t k1, uint64_t k2, uint32_t cap, uint32_t hbits, bool two = true) {
    uint32_t s1 = tab_hash(k1, hbits), s2 = tab_hash(k2, hbits);
    bool p1 = true, p2 = two;
    for (uint32_t pr = 0; pr < cap && (p1 || p2); ++pr) {
        const uint64_t v1 = lds_ld64(&tab[s1]), v2 = lds_ld64(&tab[s2]);
        if (p1) { if ((v1 >> PLO_VB) == k1) { atomicAdd((unsigned long long *)&tab[s1], ~0ull); p1 = false; } else s1 = (s1 + 1u) & (cap - 1u); }
        if (p2) { if ((v2 >> PLO_VB) == k2) { atomicAdd((unsigned long long *)&tab[s2], ~0ull); p2 = false; } else s2 = (s2 + 1u) & (cap - 1u); }
    }
    return !p1 && !p2;
}
// count[key] += 1 ; claims an empty or dead (count 0) slot for a new key.
// Only called when no decrement is in flight (see the two sweeps below), so a
// key can never end up in two slots.
__device__ __forceinline__ bool tab_inc(uint64_t *tab, uint64_t key, uint32_t cap, uint32_t hbits) {
    uint32_t s = tab_hash(key, hbits);
    for (uint32_t pr = 0; pr < 2u * cap + 64u; ++pr) {
        uint64_t v = lds_ld64(&tab[s]);
        if ((v >> PLO_VB) == key) { atomicAdd((unsigned long long *)&tab[s], 1ull); return true; }
        if ((v & PLO_VMASK) == 0ull) {
            uint64_t nv = (key << PLO_VB) | 1ull;
            uint64_t old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)nv);
            if (old == v) return true;
            continue;                      // slot changed under us: look at it again
        }
        s = (s + 1u) & (cap - 1u);
    }
    return false;
}

// generic add on the table: slot = key<<32 | value; used by ProgramGen for the
// (column, |coefficient|) multiset: low 16 bits = occurrences, bit 16 = "a
// multiplier r := t_col * |coefficient| already exists" (`multiples`).
__device__ __forceinline__ bool tab_add(uint64_t *tab, uint64_t key, uint32_t incv, uint32_t cap, uint32_t hbits) {
    uint32_t s = tab_hash(key, hbits);
    for (uint32_t pr = 0; pr < 2u * cap + 64u; ++pr) {
        uint64_t v = lds_ld64(&tab[s]);
        if (v == PLO_EMPTY) {
            uint64_t nv = (key << PLO_VB) | incv;
            uint64_t old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)nv);
            if (old == v) return true;
            continue;
        }
        if ((v >> PLO_VB) == key) { atomicAdd((unsigned long long *)&tab[s], (unsigned long long)incv); return true; }
        s = (s + 1u) & (cap - 1u);
    }
    return false;
}
// set a flag bit on the key's value (insert the key if absent); idempotent, unlike an add
__device__ __forceinline__ bool tab_flag(uint64_t *tab, uint64_t key, uint32_t flag, uint32_t cap, uint32_t hbits) {
    uint32_t s = tab_hash(key, hbits);
    for (uint32_t pr = 0; pr < 2u * cap + 64u; ++pr) {
        uint64_t v = lds_ld64(&tab[s]);
        if (v == PLO_EMPTY) {
            uint64_t old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)((key << PLO_VB) | flag));
            if (old == v) return true;
            continue;
        }
        if ((v >> PLO_VB) == key) { atomicOr((unsigned long long *)&tab[s], (unsigned long long)flag); return true; }
        s = (s + 1u) & (cap - 1u);
    }
    return false;
}
__device__ __forceinline__ uint32_t tab_find(const uint64_t *tab, uint64_t key, uint32_t cap, uint32_t hbits) {
    uint32_t s = tab_hash(key, hbits);
    for (uint32_t pr = 0; pr < cap; ++pr) {
        uint64_t v = tab[s];
        if (v == PLO_EMPTY) return 0u;
        if ((v >> PLO_VB) == key) return (uint32_t)(v & PLO_VMASK);
        s = (s + 1u) & (cap - 1u);
    }
    return 0u;
}

#define PLO_FRESH 0xFFFFu       // column id of a variable created inside ProgramGen (never looked up by index)
#define PLO_MFLAG 0x1000u       // "in multiples" flag in the table value (bit 12, above the 12-bit count)
#define PLO_PGCMASK 0xFFFu

// ProgramGen for general coefficients, counts only (reference
// include/plinopt_optimize.inl:513-611).  Counting semantics, derived from the
// reference code (DESIGN.md "ProgramGen, count-only"):
//  A FactorOutColumns :318-371  every (column j, |v|=e) with e not +-1 occurring >1 times in
//    column j costs one multiplication unless (j,e) is already in `multiples`; the entries become +-1
//    entries of a fresh column.
//  B FactorOutRows :375-420  every |v|=e not +-1 occurring f>1 times in a row costs f-1 additions and
//    collapses into one entry (fresh column, e).
//  C Triangle :427-507  scan order and the never-reset `found` flag restated literally.
//  D output rows :547-604  len-1 additions per row; one multiplication per entry whose |v| is not +-1
//    and whose (column,|v|) is not in `multiples`.
// Numbering of fresh columns never influences a count, so they all carry the id PLO_FRESH.
__device__ uint64_t program_gen_general(const WavePlan &P, uint8_t *reg, const uint16_t *rs, uint32_t lane,
                                        uint32_t ncols0, uint32_t nmult, uint32_t nbadd, uint32_t nbmul, uint32_t *errw)
{
    uint64_t *tab   = (uint64_t *)(reg + P.off_tab);
    uint64_t *cmask = (uint64_t *)(reg + P.off_cmask);
    uint64_t *umask = (uint64_t *)(reg + P.off_umask);
    uint32_t *val   = (uint32_t *)(reg + P.off_val);
    uint32_t *inv   = (uint32_t *)(reg + P.off_inv);
    uint16_t *col   = (uint16_t *)(reg + P.off_col);
    uint16_t *len   = (uint16_t *)(reg + P.off_len);
    uint32_t *multv = (uint32_t *)(reg + P.off_mult);
    uint32_t *multc = multv + P.multcap;
    const uint32_t p = P.p, cap = P.cap, hbits = P.hbits, m = P.m, ms = 2u * P.mw, rb = P.rb;
    const uint64_t mu = P.mu;
    const uint32_t LPR = 1u << P.lpr_log2, G = 64u >> P.lpr_log2;
    const uint32_t g = lane >> P.lpr_log2, t = lane & (LPR - 1u), gbase = g << P.lpr_log2;
    const uint64_t gmask = (LPR == 64u) ? ~0ull : (((1ull << LPR) - 1ull) << gbase);
    bool bad = false;

    if (P.mw != 1u) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_PGEN); return 0; }

    // multiset of (column, |v|) + the multiples set
    for (uint32_t s = lane; s < cap; s += 64u) tab[s] = PLO_EMPTY;
    PLO_WAVE_SYNC();
    for (uint32_t s0 = 0; s0 < nmult; s0 += 64u)
        if (s0 + lane < nmult) bad |= !tab_flag(tab, ((uint64_t)multc[s0 + lane] << rb) | multv[s0 + lane], PLO_MFLAG, cap, hbits);
    PLO_WAVE_SYNC();
    // A1: occurrences of (j, e)
    for (uint32_t r0 = 0; r0 < m; r0 += G) {
        const uint32_t row = r0 + g; const bool act = row < m;
        const uint32_t base = act ? rs[row] : 0u, ln = act ? len[row] : 0u;
        if (t < ln) {
            const uint32_t e = fabsp(val[base + t], p);
            if (!absone(e, p)) bad |= !tab_add(tab, ((uint64_t)col[base + t] << rb) | e, 1u, cap, hbits);
        }
    }
    PLO_WAVE_SYNC();
    // A2: one multiplication per repeated (j,e) not yet in multiples; it joins multiples
    {
        uint32_t cnt = 0;
        for (uint32_t s = lane; s < cap; s += 64u) {
            uint64_t v = tab[s];
            if (v != PLO_EMPTY && ((uint32_t)v & PLO_PGCMASK) >= 2u && !((uint32_t)v & PLO_MFLAG)) { ++cnt; tab[s] = v | PLO_MFLAG; }
        }
        nbmul += wave_sum(cnt);
    }
    PLO_WAVE_SYNC();
    // A3: repeated entries move to a fresh column with value +-1
    for (uint32_t r0 = 0; r0 < m; r0 += G) {
        const uint32_t row = r0 + g; const bool act = row < m;
        const uint32_t base = act ? rs[row] : 0u, ln = act ? len[row] : 0u;
        if (t < ln) {
            const uint32_t v = val[base + t], e = fabsp(v, p), c = col[base + t];
            if (!absone(e, p) && (tab_find(tab, ((uint64_t)c << rb) | e, cap, hbits) & PLO_PGCMASK) >= 2u) {
                const uint32_t u = (v == e) ? 1u : p - 1u;                 // Fsign >= 0 ? one : mOne
                col[base + t] = (uint16_t)PLO_FRESH; val[base + t] = u; inv[base + t] = u;
                atomicAnd((unsigned long long *)&cmask[c * ms], ~(1ull << row));
            }
        }
    }
    PLO_WAVE_SYNC();
    // B: FactorOutRows on every row
    {
        uint32_t addacc = 0;
        for (uint32_t r0 = 0; r0 < m; r0 += G) {
            const uint32_t row = r0 + g; const bool act = row < m;
            const uint32_t base = act ? rs[row] : 0u, ln = act ? len[row] : 0u;
            const bool have = t < ln;
            const uint32_t v = have ? val[base + t] : 0u, iv = have ? inv[base + t] : 0u, c = have ? col[base + t] : PLO_FRESH;
            uint32_t e = have ? fabsp(v, p) : 0u;
            if (absone(e, p)) e = 0u;
            uint32_t freq = 0, first = LPR;
            for (uint32_t u = 0; u < LPR; ++u) {
                const uint32_t eu = bcast(e, gbase + u);
                if (e != 0u && eu == e) { ++freq; if (first == LPR) first = u; }
            }
            const bool grouped = e != 0u && freq > 1u;
            const bool leader = grouped && first == t;
            if (leader) addacc += freq - 1u;
            const bool keep = have && (!grouped || leader);
            const uint64_t km = __ballot(keep) & gmask;
            if (grouped && !leader && c != PLO_FRESH) atomicAnd((unsigned long long *)&cmask[c * ms], ~(1ull << row));
            if (leader && c != PLO_FRESH) atomicAnd((unsigned long long *)&cmask[c * ms], ~(1ull << row));
            if (keep) {
                const uint32_t np = base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                if (leader) { col[np] = (uint16_t)PLO_FRESH; val[np] = e; inv[np] = (v == e) ? iv : p - iv; }
                else { col[np] = (uint16_t)c; val[np] = v; inv[np] = iv; }
            }
            if (act && t == 0u) len[row] = (uint16_t)__popcll(km);
        }
        nbadd += wave_sum(addacc);
    }
    PLO_WAVE_SYNC();
    // C: Triangle, column by column (rows on lanes; mw == 1)
    for (uint32_t j = 0; j < ncols0; ++j) {
        bool found = false;
        for (;;) {
            const uint64_t NU = uni64(cmask[j * ms] & ~umask[j * ms]);      // rows holding a non +-1 entry in column j
            if (__popcll(NU) < 2) break;
            const bool mine = (NU >> lane) & 1ull;
            uint32_t vj = 0, ivj = 0;
            if (mine) {
                const uint32_t base = rs[lane], ln = len[lane];
                for (uint32_t z = 0; z < ln; ++z) if (col[base + z] == j) { vj = val[base + z]; ivj = inv[base + z]; break; }
            }
            int it = -1, nx = -1;
            uint64_t scan = NU;
            if (found) {                                                      // only the first couple is looked at again
                const uint32_t i0 = (uint32_t)__builtin_ctzll(NU);
                scan = 1ull << i0;
            }
            while (scan) {
                const uint32_t i0 = (uint32_t)__builtin_ctzll(scan); scan &= scan - 1ull;
                const uint32_t v1 = bcast(vj, i0), iv1 = bcast(ivj, i0);
                (void)v1;
                uint64_t cand = NU & ~(1ull << i0);
                if (found) cand = 1ull << __builtin_ctzll(cand);
                bool hit = false;
                if ((cand >> lane) & 1ull) {
                    const uint32_t quot = fmul<false>(vj, iv1, p, mu), nq = p - quot;
                    const uint32_t base = rs[lane], ln = len[lane];
                    for (uint32_t z = 0; z < ln; ++z) {
                        const uint32_t tv = val[base + z];
                        if (col[base + z] != j && !absone(tv, p) && (tv == quot || tv == nq)) { hit = true; break; }
                    }
                }
                const uint64_t hm = __ballot(hit);
                if (hm) { it = (int)i0; nx = (int)__builtin_ctzll(hm); break; }
            }
            if (it < 0) break;
            // apply :453-498
            found = true;
            const uint32_t v1 = bcast(vj, (uint32_t)it), iv1 = bcast(ivj, (uint32_t)it);
            ++nbmul;                                                          // t_m := t_j * |a|
            if (lane == 0) bad |= !tab_flag(tab, ((uint64_t)j << rb) | v1, PLO_MFLAG, cap, hbits);   // multiples gets the SIGNED value (:464)
            if (lane == (uint32_t)it) {
                const uint32_t base = rs[lane], ln = len[lane];
                for (uint32_t z = 0; z < ln; ++z) if (col[base + z] == j) { col[base + z] = (uint16_t)PLO_FRESH; val[base + z] = 1u; inv[base + z] = 1u; break; }
            }
            uint32_t addone = 0;
            if (lane == (uint32_t)nx) {
                const uint32_t quot = fmul<false>(vj, iv1, p, mu), iquot = fmul<false>(ivj, v1, p, mu);
                const uint32_t eq = fabsp(quot, p), ieq = (quot == eq) ? iquot : p - iquot;
                const uint32_t base = rs[lane], ln = len[lane];
                uint32_t w = 0, f = 1;                                        // FactorOutRows on this row (:495-496)
                for (uint32_t z = 0; z < ln; ++z) {
                    const uint32_t cz = col[base + z], tv = val[base + z], ti = inv[base + z];
                    if (cz == j) continue;                                    // moved to the new column with value quot
                    if (fabsp(tv, p) == eq) { ++f; if (cz != PLO_FRESH) atomicAnd((unsigned long long *)&cmask[cz * ms], ~(1ull << lane)); continue; }
                    col[base + w] = (uint16_t)cz; val[base + w] = tv; inv[base + w] = ti; ++w;
                }
                col[base + w] = (uint16_t)PLO_FRESH; val[base + w] = eq; inv[base + w] = ieq; ++w;
                len[lane] = (uint16_t)w;
                addone = f - 1u;
            }
            nbadd += bcast(addone, (uint32_t)nx);
            if (lane == 0) cmask[j * ms] &= ~((1ull << it) | (1ull << nx));
            PLO_WAVE_SYNC();
        }
    }
    // D: output rows
    {
        uint32_t addacc = 0, mulacc = 0;
        for (uint32_t r0 = 0; r0 < m; r0 += G) {
            const uint32_t row = r0 + g; const bool act = row < m;
            const uint32_t base = act ? rs[row] : 0u, ln = act ? len[row] : 0u;
            if (act && t == 0u && ln > 1u) addacc += ln - 1u;
            if (t < ln) {
                const uint32_t e = fabsp(val[base + t], p), c = col[base + t];
                if (!absone(e, p)) {
                    const bool reuse = c != PLO_FRESH && (tab_find(tab, ((uint64_t)c << rb) | e, cap, hbits) & PLO_MFLAG);
                    if (!reuse) ++mulacc;
                }
            }
        }
        nbadd += wave_sum(addacc); nbmul += wave_sum(mulacc);
    }
    if (__ballot(bad)) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_TABLE); }
    return ((uint64_t)nbadd << 32) | nbmul;
}

// How a candidate picks among the eligible triples of a step.  Random restarts (OneSub :260-265): the triples of maximal
// frequency, one draw of the stream.  Enumeration (-E, RecSub :889-982 explores every pair of frequency > 1): all triples
// of frequency > 1 are the children, the candidate's index is read in the mixed radix of its own path (digit = rem mod T,
// rem /= T); prod = product of the radices (saturating): indices 0..N-1 cover the whole tree once N >= max prod.
struct PickState { uint32_t rng; uint32_t enumerate; uint64_t rem, prod; uint32_t recsub_muls; };   // recsub_muls: RecSub's multiplication count of the schedule (:950-951), set by run_candidate

// One candidate.  Returns packed (adds<<32 | muls); sets *errw on failure.
template <bool UNIT>
__device__ uint64_t run_candidate(const WavePlan &P, uint8_t *reg, const uint16_t *rs, PickState &ps,
                                  uint32_t lane, uint32_t *errw)
{
    uint64_t *tab   = (uint64_t *)(reg + P.off_tab);
    uint64_t *cmask = (uint64_t *)(reg + P.off_cmask);
    uint64_t *umask = (uint64_t *)(reg + P.off_umask);
    uint32_t *val   = (uint32_t *)(reg + P.off_val);
    uint32_t *inv   = (uint32_t *)(reg + P.off_inv);
    uint16_t *col   = (uint16_t *)(reg + P.off_col);
    uint16_t *len   = (uint16_t *)(reg + P.off_len);
    uint64_t *affw  = (uint64_t *)(reg + P.off_aff);      // [mw] affected rows, [mw] affected & coeff +-1, [1] scratch
    uint16_t *ties  = (uint16_t *)(reg + P.off_ties);
    uint32_t *multv = (uint32_t *)(reg + P.off_mult);     // multcap values, then multcap u32 columns
    uint32_t *multc = multv + P.multcap;

    const uint32_t p = P.p, NC = P.NC, cap = P.cap, hbits = P.hbits, mw = P.mw, ms = 2u * P.mw;
    const uint32_t rb = P.rb, bb = P.bb, abs_ = P.rb + P.bb;
#define PLO_KEY(a_, b_, r_) (((uint64_t)(a_) << abs_) | ((uint64_t)(b_) << rb) | (uint64_t)(r_))   // masks interleaved: {cmask[mw],umask[mw]} per column
    const uint64_t mu = P.mu;
    const uint32_t LPR = 1u << P.lpr_log2, G = 64u >> P.lpr_log2;
    const uint32_t g = lane >> P.lpr_log2, t = lane & (LPR - 1u), gbase = g << P.lpr_log2;
    const uint64_t gmask = (LPR == 64u) ? ~0ull : (((1ull << LPR) - 1ull) << gbase);

    uint32_t ncols = P.n, nbadd = 0, nbmul = 0, nmult = 0;

#ifdef PLO_WAVE_PROFILE
    unsigned long long wp_t = clock64(), wp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, wp_steps = 0, wp_s = 0;
#endif
    for (;;) {
        if (ncols >= NC) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_STEPS); break; }
        // ---- OneSub :244-253  maximal frequency over the pair table
        uint32_t lmax = 0;
        for (uint32_t s = lane; s < cap; s += 64u) { uint32_t c = (uint32_t)(tab[s] & PLO_VMASK); lmax = c > lmax ? c : lmax; }
        const uint32_t maxfrq = uni32(wave_max(lmax));
        WP_T(0);
        if (maxfrq <= 1u) break;                                            // :255
        // ---- ties in map order; random pick :260-265
        uint32_t T = 0;
        for (uint32_t s0 = 0; s0 < cap; s0 += 64u) {
            uint64_t v = tab[s0 + lane];
            const uint32_t cnt_ = (uint32_t)(v & PLO_VMASK);
            bool is = ps.enumerate ? cnt_ >= 2u : cnt_ == maxfrq;
            uint64_t bm = __ballot(is);
            if (is) ties[T + __popcll(bm & ((1ull << lane) - 1ull))] = (uint16_t)(s0 + lane);
            T += (uint32_t)__popcll(bm);
        }
        T = uni32(T);
        PLO_WAVE_SYNC();
        WP_T(1);
        uint64_t key;
        if (T == 1u) {
            key = tab[ties[0]] >> PLO_VB;
        } else {
            uint32_t k;
            if (ps.enumerate) {
                k = (uint32_t)(ps.rem % T); ps.rem /= T;
                ps.prod = ps.prod > 0xFFFFFFFFFFFFFFFFull / T ? 0xFFFFFFFFFFFFFFFFull : ps.prod * T;
            } else k = rng_next(ps.rng) % T;
            k = uni32(k);
            if (T <= 64u) {
                const uint64_t mine = lane < T ? (tab[ties[lane]] >> PLO_VB) : ~0ull;
                uint32_t rank = 0;
                for (uint32_t j = 0; j < T; ++j) rank += (rdlane64(mine, j) < mine) ? 1u : 0u;      // j is wave-uniform: v_readlane, not an LDS permute
                uint64_t w = __ballot(lane < T && rank == k);
                key = rdlane64(mine, (uint32_t)__builtin_ctzll(w));
            } else {
                uint64_t lo = 0ull, hi = (1ull << (64u - PLO_VB)) - 1ull;  // k-th smallest by bisection on the key value (keys have at most 64 - PLO_VB bits)
                while (lo < hi) {
                    uint64_t mid = lo + ((hi - lo) >> 1); uint32_t c = 0;
                    for (uint32_t s0 = 0; s0 < T; s0 += 64u) {
                        bool le = (s0 + lane < T) && (tab[ties[s0 + lane]] >> PLO_VB) <= mid;
                        c += (uint32_t)__popcll(__ballot(le));
                    }
                    if (c >= k + 1u) hi = mid; else lo = mid + 1ull;
                }
                key = lo;
            }
        }
        key = uni64(key);
        WP_T(2);
        ++nbadd;                                                            // :292
        // ---- RemOneCSE :60-194
        const uint32_t r = (uint32_t)(key & ((1ull << rb) - 1ull)), b = (uint32_t)(key >> rb) & ((1u << bb) - 1u), a = (uint32_t)(key >> abs_);
        uint32_t c0 = 0, c1 = 0;
        for (uint32_t w = 0; w < mw; ++w) { c0 += (uint32_t)__popcll(umask[a * ms + w]); c1 += (uint32_t)__popcll(umask[b * ms + w]); }
        const bool swap = uni32(c0) < uni32(c1);                                          // :79-88
        const uint32_t l0 = swap ? b : a, l1 = swap ? a : b, lm = ncols;
        if (lane < 2u * mw + 1u) affw[lane] = 0ull;
        PLO_WAVE_SYNC();
        bool bad = false;
        // Most steps touch at most G rows (G = lane groups of a wave): one trip does everything.  The lanes of a group keep their
        // row's entries in registers across the barrier that separates the retirements (:115-118) from the insertions (:132-142)
        // -- the table rule "no increment while a decrement is in flight" still holds -- so the row is read, searched for the two
        // columns and broadcast once, not once per sweep.
        const uint64_t msk0 = mw == 1u ? uni64(cmask[a * ms] & cmask[b * ms]) : 0ull;
        if (mw == 1u && (uint32_t)__popcll(msk0) <= G) {
            const uint32_t pc = (uint32_t)__popcll(msk0);
            const int myrow = g < pc ? (int)nth_set_bit(msk0, g) : -1;
            const bool act = myrow >= 0;
            const uint32_t base = act ? rs[myrow] : 0u, ln = act ? len[myrow] : 0u;
            const bool have = t < ln;
            const uint32_t c = have ? col[base + t] : 0xFFFFu, v = have ? val[base + t] : 0u;
            const uint32_t iv = UNIT ? v : (have ? inv[base + t] : 0u);
            const uint64_t ma = __ballot(have && c == a) & gmask, mb = __ballot(have && c == b) & gmask;
            const uint32_t la = ma ? (uint32_t)__builtin_ctzll(ma) : lane, lb = mb ? (uint32_t)__builtin_ctzll(mb) : lane;
            const uint32_t va = bcast(v, la), ia = bcast(iv, la), vb = bcast(v, lb), ib = bcast(iv, lb);
            const bool aff = act && ma && mb && vb == fmul<UNIT>(r, va, p, mu);
            if (aff && have && lane != lb) {
                const bool isa = lane == la;
                const uint64_t k1 = isa ? key : (c < a ? PLO_KEY(c, a, fmul<UNIT>(va, iv, p, mu)) : PLO_KEY(a, c, fmul<UNIT>(v, ia, p, mu)));
                const uint64_t k2 = c < b ? PLO_KEY(c, b, fmul<UNIT>(vb, iv, p, mu)) : PLO_KEY(b, c, fmul<UNIT>(v, ib, p, mu));
                bad |= !tab_dec2(tab, k1, k2, cap, hbits, !isa);
                if (isa) {
                    atomicOr((unsigned long long *)&affw[0], 1ull << (uint32_t)myrow);
                    if (!UNIT) affw[2u] = fmul<UNIT>(va, ib, p, mu);           // 1/r, same in every affected row
                }
            }
            PLO_WAVE_SYNC();
            WP_T(3);
            if (aff && have) {
                const bool first = l0 == a;                                     // the entry of l0 carries the new column's coefficient
                const uint32_t q0 = first ? la : lb, q1 = first ? lb : la;
                const uint32_t coeff = first ? va : vb, icoeff = first ? ia : ib;
                if (lane != q0 && lane != q1) {
                    bad |= !tab_inc(tab, PLO_KEY(c, lm, fmul<UNIT>(coeff, iv, p, mu)), cap, hbits);
                    const uint32_t np = base + t - (lane > q0 ? 1u : 0u) - (lane > q1 ? 1u : 0u);
                    col[np] = (uint16_t)c; val[np] = v; if (!UNIT) inv[np] = iv;
                } else if (lane == q0) {
                    const uint32_t np = base + ln - 2u;
                    col[np] = (uint16_t)lm; val[np] = coeff; if (!UNIT) inv[np] = icoeff;
                    len[myrow] = (uint16_t)(ln - 1u);
                    if (UNIT || absone(coeff, p)) atomicOr((unsigned long long *)&affw[1], 1ull << (uint32_t)myrow);
                }
            }
        } else {
            // sweep 1: rows holding the triple (a,b,r); retire their old pairs (:115-118)
            for (uint32_t w = 0; w < mw; ++w) {
                uint64_t msk = uni64(cmask[a * ms + w] & cmask[b * ms + w]);
                while (msk) {
                    // group g takes the g-th row of the mask (bit arithmetic per lane: no scalar loop over the groups)
                    const uint32_t pc = (uint32_t)__popcll(msk);
                    const int myrow = g < pc ? (int)(w * 64u + nth_set_bit(msk, g)) : -1;
                    if (pc <= G) msk = 0ull;
                    else msk &= ~((2ull << ((uint32_t)__builtin_amdgcn_readlane(myrow, (int)((G - 1u) << P.lpr_log2)) & 63u)) - 1ull);
                    const bool act = myrow >= 0;
                    const uint32_t base = act ? rs[myrow] : 0u, ln = act ? len[myrow] : 0u;
                    const bool have = t < ln;
                    const uint32_t c = have ? col[base + t] : 0xFFFFu, v = have ? val[base + t] : 0u;
                    const uint32_t iv = UNIT ? v : (have ? inv[base + t] : 0u);
                    const uint64_t ma = __ballot(have && c == a) & gmask, mb = __ballot(have && c == b) & gmask;
                    const uint32_t la = ma ? (uint32_t)__builtin_ctzll(ma) : lane, lb = mb ? (uint32_t)__builtin_ctzll(mb) : lane;
                    const uint32_t va = bcast(v, la), ia = bcast(iv, la), vb = bcast(v, lb), ib = bcast(iv, lb);
                    const bool aff = act && ma && mb && vb == fmul<UNIT>(r, va, p, mu);
#ifdef PLO_WAVE_PROFILE
                    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t__ = clock64(); wp_acc[6] += t__ - wp_t; wp_s = t__; ++wp_acc[9]; }
#endif
                    if (aff && have && lane != lb) {
                        // one probe loop for every lane: the entry of a retires the chosen triple itself, the others their pairs with a and with b
                        const bool isa = lane == la;
                        const uint64_t k1 = isa ? key : (c < a ? PLO_KEY(c, a, fmul<UNIT>(va, iv, p, mu)) : PLO_KEY(a, c, fmul<UNIT>(v, ia, p, mu)));
                        const uint64_t k2 = c < b ? PLO_KEY(c, b, fmul<UNIT>(vb, iv, p, mu)) : PLO_KEY(b, c, fmul<UNIT>(v, ib, p, mu));
                        bad |= !tab_dec2(tab, k1, k2, cap, hbits, !isa);
                        if (isa) {
                            atomicOr((unsigned long long *)&affw[(uint32_t)myrow >> 6], 1ull << ((uint32_t)myrow & 63u));
                            if (!UNIT) affw[2u * mw] = fmul<UNIT>(va, ib, p, mu);   // 1/r, same in every affected row
                        }
                    }
#ifdef PLO_WAVE_PROFILE
                    { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t__ = clock64(); wp_acc[7] += t__ - wp_s; wp_t = t__; }
#endif
                }
            }
            PLO_WAVE_SYNC();
            WP_T(3);
            // sweep 2: rewrite the rows, add the pairs with the new column (:96-110, :132-142)
            for (uint32_t w = 0; w < mw; ++w) {
                uint64_t msk = uni64(affw[w]);
                while (msk) {
                    // group g takes the g-th row of the mask (bit arithmetic per lane: no scalar loop over the groups)
                    const uint32_t pc = (uint32_t)__popcll(msk);
                    const int myrow = g < pc ? (int)(w * 64u + nth_set_bit(msk, g)) : -1;
                    if (pc <= G) msk = 0ull;
                    else msk &= ~((2ull << ((uint32_t)__builtin_amdgcn_readlane(myrow, (int)((G - 1u) << P.lpr_log2)) & 63u)) - 1ull);
                    const bool act = myrow >= 0;
                    const uint32_t base = act ? rs[myrow] : 0u, ln = act ? len[myrow] : 0u;
                    const bool have = t < ln;
                    const uint32_t c = have ? col[base + t] : 0xFFFFu, v = have ? val[base + t] : 0u;
                    const uint32_t iv = UNIT ? v : (have ? inv[base + t] : 0u);
                    const uint64_t m0 = __ballot(have && c == l0) & gmask, m1 = __ballot(have && c == l1) & gmask;
                    const uint32_t q0 = m0 ? (uint32_t)__builtin_ctzll(m0) : lane, q1 = m1 ? (uint32_t)__builtin_ctzll(m1) : lane;
                    const uint32_t coeff = bcast(v, q0), icoeff = bcast(iv, q0);
                    if (have) {
                        if (lane != q0 && lane != q1) {
                            bad |= !tab_inc(tab, PLO_KEY(c, lm, fmul<UNIT>(coeff, iv, p, mu)), cap, hbits);
                            const uint32_t np = base + t - (lane > q0 ? 1u : 0u) - (lane > q1 ? 1u : 0u);
                            col[np] = (uint16_t)c; val[np] = v; if (!UNIT) inv[np] = iv;
                        } else if (lane == q0) {
                            const uint32_t np = base + ln - 2u;
                            col[np] = (uint16_t)lm; val[np] = coeff; if (!UNIT) inv[np] = icoeff;
                            len[myrow] = (uint16_t)(ln - 1u);
                            if (UNIT || absone(coeff, p)) atomicOr((unsigned long long *)&affw[mw + ((uint32_t)myrow >> 6)], 1ull << ((uint32_t)myrow & 63u));
                        }
                    }
                }
            }
        }
        PLO_WAVE_SYNC();
        WP_T(4);
        if (lane < mw) {
            const uint64_t af = affw[lane], uaf = affw[mw + lane];
            cmask[l0 * ms + lane] &= ~af; cmask[l1 * ms + lane] &= ~af;
            umask[l0 * ms + lane] &= ~af; umask[l1 * ms + lane] &= ~af;
            cmask[lm * ms + lane] = af;   umask[lm * ms + lane] = uaf;
        }
        if (__ballot(bad)) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_TABLE); break; }
        // ---- multiplier reuse :153-169 (counts only)
        if (!UNIT) {
            const uint32_t rho = swap ? (uint32_t)affw[2u * mw] : r;
            const uint32_t asgs = fabsp(rho, p);
            if (!absone(asgs, p)) {
                bool hit = false;
                for (uint32_t s0 = 0; s0 < nmult; s0 += 64u)
                    hit |= (s0 + lane < nmult) && multc[s0 + lane] == l1 && multv[s0 + lane] == asgs;
                if (!__ballot(hit)) {
                    if (nmult >= P.multcap) { if (lane == 0) atomicMax(errw, (uint32_t)ERR_MULT); break; }
                    if (lane == 0) { multv[nmult] = asgs; multc[nmult] = l1; }
                    ++nmult; ++nbmul;
                }
            }
        }
        ncols = lm + 1u;                                                    // :190-191
        PLO_WAVE_SYNC();
        WP_T(5);
#ifdef PLO_WAVE_PROFILE
        ++wp_steps;
#endif
    }
#ifdef PLO_WAVE_PROFILE
    if (lane == 0) { for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&g_wprof[q_], wp_acc[q_]); atomicAdd(&g_wprof[8], wp_steps); atomicAdd(&g_wprof[6], wp_acc[6]); atomicAdd(&g_wprof[7], wp_acc[7]); atomicAdd(&g_wprof[10], wp_acc[8]); atomicAdd(&g_wprof[11], wp_acc[9]); }
#endif

    // ---- ProgramGen :513-611
    if (UNIT) {
        // all coefficients are +-1: the three factoring passes are no-ops and
        // every row costs len-1 additions (:576), no multiplication
        uint32_t acc = 0;
        for (uint32_t i = lane; i < P.m; i += 64u) { uint32_t ln = len[i]; acc += ln > 1u ? ln - 1u : 0u; }
        nbadd += wave_sum(acc);
        ps.recsub_muls = 0u;
    } else {
        PLO_WAVE_SYNC();
        {   // RecSub counts, per schedule, the multipliers emitted so far plus one multiplication per non +-1 entry left (:950-951, :1001-1003)
            uint32_t nu = 0;
            for (uint32_t i = lane; i < P.m; i += 64u) { const uint32_t b0 = rs[i], ln = len[i]; for (uint32_t k = 0; k < ln; ++k) { const uint32_t v = val[b0 + k]; nu += (v != 1u && v != p - 1u) ? 1u : 0u; } }
            ps.recsub_muls = nbmul + wave_sum(nu);
        }
        return program_gen_general(P, reg, rs, lane, ncols, nmult, nbadd, nbmul, errw);
    }
    return ((uint64_t)nbadd << 32) | nbmul;
}

__device__ __forceinline__ uint32_t cost_key32(uint32_t a, uint32_t mu_, uint32_t mode) {
    switch (mode) {
    case 1: return (a << 16) | mu_;              // adds, then muls
    case 2: return (a + mu_) << 16;              // sum only
    case 3: return (a << 16) | mu_;              // RecSub's order: adds, then ITS multiplication count (the caller passes it as mu_)
    default: return ((a + mu_) << 16) | a;       // sum, then adds
    }
}

template <bool UNIT>
__global__ __launch_bounds__(256, UNIT ? PLO_WAVE_OCC_UNIT : PLO_WAVE_OCC_GEN) void cse_wave_kernel(WavePlan P, WaveJob J)
{
    extern __shared__ uint64_t lds64[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // shared: rs[m+1] (row offsets never change); per wave: one candidate region
    const uint32_t rs_words = P.rs_bytes >> 3, tw = P.tmpl_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < rs_words; i += blockDim.x) lds64[i] = P.tmpl[tw + i];
    __syncthreads();
    const uint16_t *rs = (const uint16_t *)lds64;
    uint8_t *reg = (uint8_t *)lds64 + P.rs_bytes + (size_t)wave * P.region_bytes;
    uint64_t best = ~0ull;
    const uint64_t stride = (uint64_t)gridDim.x * nwaves;
    for (uint64_t c = (uint64_t)blockIdx.x * nwaves + wave; c < J.ncand; c += stride) {
        for (uint32_t i = lane; i < tw; i += 64u) ((uint64_t *)reg)[i] = P.tmpl[i];     // matrix image -> LDS
        PLO_WAVE_SYNC();
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        PickState ps{1u + (uint32_t)(splitmix64(seed) % 2147483646ull), J.enumerate, seed, 1ull, 0u};
        const uint64_t res = run_candidate<UNIT>(P, reg, rs, ps, lane, J.err);
#ifdef PLO_WAVE_PROFILE
        if (lane == 0) atomicAdd(&g_wprof[9], 1ull);
#endif
        if (J.enumerate && lane == 0) { if (J.prods) J.prods[c] = ps.prod; if (J.prodmax) atomicMax(J.prodmax, (unsigned long long)ps.prod); }
        const uint32_t a = (uint32_t)(res >> 32), mu_ = (uint32_t)res;
        if (lane == 0) {
            if (J.adds) J.adds[c] = a;
            if (J.muls) J.muls[c] = mu_;
        }
        const uint64_t packed = ((uint64_t)cost_key32(a, J.cost_mode == 3u ? ps.recsub_muls : mu_, J.cost_mode) << 32) | (uint32_t)c;
        best = packed < best ? packed : best;
        PLO_WAVE_SYNC();
    }
    if (J.best) {
        // grid min-reduce: wave value is uniform; waves -> LDS -> one atomicMin per workgroup
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

// Two matrices per candidate, one random stream: the restart loop of LUOptimiser (reference
// include/plinopt_optimize.inl:1056-1100) runs Optimizer() on a copy of U and then on a copy of L and adds
// the two op-counts (:1068-1079); the thread's generator simply keeps running from one call to the next.
__global__ __launch_bounds__(256) void cse_chain_kernel(WavePlan P1, WavePlan P2, WaveJob J)
{
    extern __shared__ uint64_t lds64[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const uint32_t rw1 = P1.rs_bytes >> 3, rw2 = P2.rs_bytes >> 3, tw1 = P1.tmpl_bytes >> 3, tw2 = P2.tmpl_bytes >> 3;
    for (uint32_t i = threadIdx.x; i < rw1; i += blockDim.x) lds64[i] = P1.tmpl[tw1 + i];
    for (uint32_t i = threadIdx.x; i < rw2; i += blockDim.x) lds64[rw1 + i] = P2.tmpl[tw2 + i];
    __syncthreads();
    const uint16_t *rs1 = (const uint16_t *)lds64, *rs2 = (const uint16_t *)(lds64 + rw1);
    const uint32_t region = P1.region_bytes > P2.region_bytes ? P1.region_bytes : P2.region_bytes;
    uint8_t *reg = (uint8_t *)lds64 + P1.rs_bytes + P2.rs_bytes + (size_t)wave * region;
    uint64_t best = ~0ull;
    const uint64_t stride = (uint64_t)gridDim.x * nwaves;
    for (uint64_t c = (uint64_t)blockIdx.x * nwaves + wave; c < J.ncand; c += stride) {
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        PickState ps{1u + (uint32_t)(splitmix64(seed) % 2147483646ull), 0u, 0ull, 1ull, 0u};
        for (uint32_t i = lane; i < tw1; i += 64u) ((uint64_t *)reg)[i] = P1.tmpl[i];
        PLO_WAVE_SYNC();
        const uint64_t r1 = P1.unit ? run_candidate<true>(P1, reg, rs1, ps, lane, J.err) : run_candidate<false>(P1, reg, rs1, ps, lane, J.err);
        PLO_WAVE_SYNC();
        for (uint32_t i = lane; i < tw2; i += 64u) ((uint64_t *)reg)[i] = P2.tmpl[i];
        PLO_WAVE_SYNC();
        const uint64_t r2 = P2.unit ? run_candidate<true>(P2, reg, rs2, ps, lane, J.err) : run_candidate<false>(P2, reg, rs2, ps, lane, J.err);
        const uint32_t a = (uint32_t)(r1 >> 32) + (uint32_t)(r2 >> 32), mu_ = (uint32_t)r1 + (uint32_t)r2;
        if (lane == 0) {
            if (J.adds) J.adds[c] = a;
            if (J.muls) J.muls[c] = mu_;
        }
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

template __global__ void cse_wave_kernel<true>(WavePlan, WaveJob);
template __global__ void cse_wave_kernel<false>(WavePlan, WaveJob);


// The same chain for MANY pairs of matrices in one launch: candidate c runs on pair c / per (the kernel method draws a
// decomposition -- hence a pair (Free, Dep) -- per restart or per small block of restarts; one launch per pair leaves the
// GPU idle).  plans[2q], plans[2q+1] are the plans of pair q in global memory; their templates carry the row starts behind
// the image, which each wave stages in its own LDS region because the pair changes with the candidate.
__global__ __launch_bounds__(256) void cse_chain_batch_kernel(const WavePlan *plans, uint32_t per, uint32_t region, uint32_t rsmax, WaveJob J)
{
    extern __shared__ uint64_t lds64[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint8_t *base = (uint8_t *)lds64 + 64u + (size_t)wave * (2u * rsmax + region);      // first 64 bytes: the reduction scratch
    uint16_t *rs1 = (uint16_t *)base, *rs2 = (uint16_t *)(base + rsmax);
    uint8_t *reg = base + 2u * rsmax;
    uint64_t best = ~0ull;
    const uint64_t stride = (uint64_t)gridDim.x * nwaves;
    for (uint64_t c = (uint64_t)blockIdx.x * nwaves + wave; c < J.ncand; c += stride) {
        const uint64_t q = c / per;
        const WavePlan &P1 = plans[2u * q], &P2 = plans[2u * q + 1u];
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        PickState ps{1u + (uint32_t)(splitmix64(seed) % 2147483646ull), 0u, 0ull, 1ull, 0u};
        const uint32_t tw1 = P1.tmpl_bytes >> 3, tw2 = P2.tmpl_bytes >> 3, rw1 = P1.rs_bytes >> 3, rw2 = P2.rs_bytes >> 3;
        for (uint32_t i = lane; i < rw1; i += 64u) ((uint64_t *)rs1)[i] = P1.tmpl[tw1 + i];
        for (uint32_t i = lane; i < rw2; i += 64u) ((uint64_t *)rs2)[i] = P2.tmpl[tw2 + i];
        for (uint32_t i = lane; i < tw1; i += 64u) ((uint64_t *)reg)[i] = P1.tmpl[i];
        PLO_WAVE_SYNC();
        const uint64_t r1 = P1.unit ? run_candidate<true>(P1, reg, rs1, ps, lane, J.err) : run_candidate<false>(P1, reg, rs1, ps, lane, J.err);
        PLO_WAVE_SYNC();
        for (uint32_t i = lane; i < tw2; i += 64u) ((uint64_t *)reg)[i] = P2.tmpl[i];
        PLO_WAVE_SYNC();
        const uint64_t r2 = P2.unit ? run_candidate<true>(P2, reg, rs2, ps, lane, J.err) : run_candidate<false>(P2, reg, rs2, ps, lane, J.err);
        const uint32_t a = (uint32_t)(r1 >> 32) + (uint32_t)(r2 >> 32), mu_ = (uint32_t)r1 + (uint32_t)r2;
        if (lane == 0) {
            if (J.adds) J.adds[c] = a;
            if (J.muls) J.muls[c] = mu_;
        }
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

} // namespace plo
