// ===========================================================================
// plo_cse_big.hip -- gfx950 kernel family for candidates that do NOT fit LDS
// (BASELINE config 5: 32x32x32_15096_L = 15096x1024, 1.26 M non-zeros, 3.15 M
// distinct pair triples, ~6.8 k CSE steps per candidate).  One WORKGROUP per
// candidate, candidate state resident in HBM, workgroups persistent over the
// seed range.  Same per-candidate semantics as plo_cse_wave.hip (reference
// include/plinopt_optimize.inl:616-631), different bookkeeping:
//
//   tab[cap]   u64  global open-addressing pair table, key48<<16 | count16 with
//                   key = col_a<<(bb+rb) | col_b<<rb | ratio: integer order ==
//                   std::map order of (size_t,size_t,Element); the initial image
//                   is built once on the host and copied per candidate.
//   rows       ONE u32 per entry at fixed row offsets rs[i] (rows only shrink):
//              column (15 bits) | +-1 flag (bit 15) | value index (16 bits).  A CSE step
//              never creates a value: the new column's entry carries the value of one of
//              the two entries it replaces, so the values of a candidate are the input's
//              distinct values (25 on config 5); {value, inverse} per index live in LDS
//              (global memory above 512 values).  ProgramGen, which does create values,
//              first expands the rows to col/val/inv arrays.
//   pruning    a triple of frequency 1 can never be chosen (OneSub :255 stops at 1) and
//              frequencies only fall after the step that creates a triple, so triples
//              are kept in the table only from frequency 2 on: count-1 triples of the
//              input are not in the image, a fresh triple seen once in its step is not
//              inserted, and retiring an absent triple is a no-op.
//   levels     the maximal frequency M never increases during a candidate, and
//              only ~60 distinct values occur on config 5.  hist[f] (LDS) counts
//              triples per frequency; cntM[c] counts triples of frequency M whose
//              first column is c; DM lists the keys that reached M; HL lists keys
//              with frequency >= theta (a window of levels) so that a level change
//              re-reads HL instead of the 16 MB table.
//   tie pick   prefix over cntM -> first column; the few DM keys of that column
//              are sorted in LDS -> k-th tie in map order (OneSub :244-265).
//   rows(a)    row lists per column (a per-candidate copy of the static transpose for input
//              columns, an append-only pool for created columns).  A step walks the SHORTER
//              list of its two columns; the rows of that list that no longer hold the column,
//              or lose it in this step, are dropped from it on the way (round 4: the lists
//              used to keep every stale row -- 2.26e6 rows searched per candidate on config 5
//              for 1.21e6 rows rewritten); +-1 counts per column kept incrementally (:70-77).
//   aggregate  the rows rewritten by one step retire/create the same triples ~21
//              times: an LDS table of 2^13 (column, ratio) entries sums them per
//              sweep; both retirements of an entry (its pair with a and with b)
//              share one LDS entry because v_b = r v_a in every affected row.  With at
//              most 32 distinct values (mode 2) an entry is a 25-bit (column, ratio
//              identifier) key + 16-bit count and the slot list covers every slot.
//   flush      pass 1 (retirements): every lane walks its share of the slot list on
//              its own (next entry as soon as both keys are settled); every table slot
//              has ONE writer in this pass, so the new frequency is a plain store of
//              the probed word.  Pass 2 (insertions, after a barrier): CAS claims.
//   scope      a candidate never leaves its workgroup: all atomics, atomic loads
//              and fences on its workspace are WORKGROUP scope, so the XCD's L2
//              serves them (agent scope = memory side of the fabric on a part whose
//              XCD L2s are not coherent with each other: 2.3x slower).
// The work per candidate is ~2.5e8 pair retirements + 1.2e8 pair insertions (1.5e7 table
// updates after aggregation).  Round 4 state: 5.8e7 L2 requests and 1.85 GB of HBM traffic per
// candidate (30x the algorithmic bytes), and NOT bound by the memory system: a candidate alone
// on the chip takes 0.44 s, 512 together 0.63 s each -- the bound is the dependent chain of one
// workgroup per candidate, two per CU (round 2's "random-access request rate"
// diagnosis was retracted in round 3).
// ===========================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plo {

#ifndef PLO_BIG_THREADS
#define PLO_BIG_THREADS 512
#endif
#define PLO_BIG_SELCAP 512u
#define PLO_GVB 16u
#define PLO_GVMASK 0xFFFFull
#define PLO_GEMPTY 0xFFFFFFFFFFFF0000ull

struct BigPlan {
    uint32_t m, n, nnz, p, NCmax, hbits, rb, bb, unit, maxf0, M0, multcap, dmcap, hlcap, scr_stride, aggbits, mers, agg_cb, agg_dual, selcap, prune;   // agg_cb: count bits of an LDS aggregation entry; agg_dual: its key is (column, x, 1/x)
    uint64_t mu;
    uint32_t nv, vt_lds;                  // distinct values; their {value, inverse} table is staged in LDS when it has <= 512 entries
    uint32_t mode, nr;                    // kernel instance: 0 value table in global memory, 1 in LDS, 2 ratio identifiers (nv <= 32); nr = distinct ratios v_i/v_j
    const uint16_t *rtid, *invid;         // mode 2: identifier of v_i/v_j at [i*32+j] (PLO_RSTRIDE); identifier of the inverse ratio
    const uint32_t *rval;                 // mode 2: the ratio of an identifier
    const uint32_t *rs, *ent0, *tptr, *trows, *ucount0, *hist0;
    const uint2 *vt;                      // {value, inverse} per value index
    const uint32_t *invtab;               // 1/x for every residue x (p <= 2^20), or nullptr
    const uint64_t *tab0;
    uint8_t *ws; uint64_t ws_stride;
    uint64_t o_tab, o_ent, o_col, o_val, o_inv, o_len, o_ucount, o_cntM, o_dm, o_hl, o_aff, o_ncrptr, o_ncr, o_multc, o_multv,
             o_tcnt, o_tptr2, o_tlist, o_cols2, o_spill,
             o_tl, o_clen, o_keep;            // per-candidate row lists of the input columns (compacted as they are walked), live length per column, scratch
    // deferred cold updates (DEFER, see "deferred" below): hot table + partitioned store + update log instead of one big table
    uint32_t defer, pbits, capp, plcap, logcap, logtrig, hwin, hotbits_min, hotbits_max, lgrp;
    uint32_t fwin;                        // entries per window of the flat sweep (2048 = 32 trips: the marks of a window are 32 words per wave; a test knob makes it smaller)
    const uint64_t *st0; const uint32_t *pcount0;     // store image: 2^pbits partitions of capp entries (key48<<16 | count16), entries per partition
    uint64_t o_store, o_pcount, o_ptail, o_log, o_plog, o_hot;
    uint32_t kb, idk;                     // idk: the ratio field of a pair key holds the ratio's identifier (kb bits), not the residue (cse_big_kernel<2, ., true>)
    // -DPLO_BIG_DIRECT (an experiment of round 4) -- mode 2 with deferred updates: columns of the DIRECT count table of the flat
    // sweep (0: none), identifiers of the ratios 1 and -1 (0xFFFF: -1 is no ratio of the matrix), value index of -v for every value index (0xFF: none)
    uint32_t dcols, id_one, id_mone;
    const uint8_t *negidx;
};

struct BigJob {
    uint64_t seed0; const uint64_t *seeds; uint64_t ncand;
    uint32_t *adds, *muls; unsigned long long *best; uint32_t cost_mode; uint32_t *err;
    unsigned long long *next;     // work counter: candidates are handed out dynamically
    uint32_t *stats;              // optional: [0]=steps of last candidate, [1]=full scans, [2]=level rebuilds
};

enum { BERR_TABLE = 11, BERR_FREQ = 12, BERR_COLS = 13, BERR_DM = 14, BERR_HL = 15, BERR_MULT = 16, BERR_SEL = 17, BERR_PGEN = 18 };

// A candidate lives in one workgroup: every atomic, atomic load and fence on its workspace is workgroup scope.  On
// gfx950 agent-scope atomics and sc1 loads are performed at the memory side of the fabric (the 8 XCDs' L2s are not
// coherent with each other) -- measured: 1.8e7 DRAM atomics per candidate before this change -- while workgroup scope
// lets the XCD's L2 do them.  Only the work counter, the error word and the best word are shared between workgroups.
#ifdef PLO_BIG_AGENT_SCOPE
#define PLO_BIG_SCOPE __HIP_MEMORY_SCOPE_AGENT
#define PLO_BIG_FENCE() __threadfence()
#else
#define PLO_BIG_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#define PLO_BIG_FENCE() __threadfence_block()
#endif
template <class T> __device__ __forceinline__ T wg_add(T *p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
template <class T> __device__ __forceinline__ T wg_sub(T *p, T v) { return __hip_atomic_fetch_add(p, (T)(0 - v), __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
template <class T> __device__ __forceinline__ T wg_or(T *p, T v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
template <class T> __device__ __forceinline__ T wg_max(T *p, T v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
template <class T> __device__ __forceinline__ T wg_cas(T *p, T expected, T desired) {
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, PLO_BIG_SCOPE);
    return expected;
}
__device__ __forceinline__ uint64_t gload64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
__device__ __forceinline__ void gstore64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
__device__ __forceinline__ uint32_t gload32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, PLO_BIG_SCOPE); }
__device__ __forceinline__ uint32_t ghash(uint64_t key, uint32_t hbits) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> (64u - hbits));
}
__device__ __forceinline__ uint32_t bmul(uint32_t a, uint32_t b, uint32_t p, uint64_t mu, uint32_t mers) {
    uint64_t x = (uint64_t)a * b;
    if (mers) {                                   // p = 2^mers - 1 (e.g. 131071 = 2^17 - 1): two folds and a correction
        x = (x & p) + (x >> mers); x = (x & p) + (x >> mers);
        return (uint32_t)(x >= p ? x - p : x);
    }
    uint64_t q = __umul64hi(x, mu);
    uint64_t r = x - q * p;                       // Barrett: at most 2p too large
    r -= r >= p ? p : 0ull; r -= r >= p ? p : 0ull;
    return (uint32_t)r;
}
// 1/x mod p (x != 0): x^(p-2)
__device__ __forceinline__ uint32_t binv(uint32_t x, uint32_t p, uint64_t mu, uint32_t mers) {
    uint32_t res = 1u, bs = x;
    for (uint32_t e = p - 2u; e; e >>= 1) { if (e & 1u) res = bmul(res, bs, p, mu, mers); bs = bmul(bs, bs, p, mu, mers); }
    return res;
}
__device__ __forceinline__ bool babsone(uint32_t e, uint32_t p) { return e == 1u || e == p - 1u; }
__device__ __forceinline__ uint32_t babs(uint32_t e, uint32_t p) { uint32_t a = e ? p - e : 0u; return a < e ? a : e; }

// An insertion that has not found its key, an empty or a dead slot after this many probes reports the table full (the callers set
// BERR_TABLE, the host rebuilds the plan with four times the slots): at the loads the table is planned for no probe sequence comes
// near it, and a table that is really full -- a candidate whose live triples outgrow the input's (tests/soak_hbm.py) -- must not cost
// 2^22 dependent memory round trips per key (seconds each: the launch looked hung).  Lookups end at the first empty slot.
#define PLO_GPROBE_INS 16384u
// frequency[key] -= 1; returns the frequency before (0 = key not found: corruption)
__device__ __forceinline__ uint32_t gtab_dec(uint64_t *tab, uint64_t key, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < (1u << 22); ++pr) {
        uint64_t v = gload64(&tab[s]);
        if ((v >> PLO_GVB) == key) { uint64_t old = wg_add((unsigned long long *)&tab[s], ~0ull); return (uint32_t)(old & PLO_GVMASK); }
        if (v == PLO_GEMPTY) return 0u;
        s = (s + 1u) & mask;
    }
    return 0u;
}
// frequency[key] += 1 (claims an empty or dead slot); returns the new frequency (0 = table full)
__device__ __forceinline__ uint32_t gtab_inc(uint64_t *tab, uint64_t key, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < PLO_GPROBE_INS; ++pr) {
        uint64_t v = gload64(&tab[s]);
        if ((v >> PLO_GVB) == key) { uint64_t old = wg_add((unsigned long long *)&tab[s], 1ull); return (uint32_t)(old & PLO_GVMASK) + 1u; }
        if ((v & PLO_GVMASK) == 0ull) {
            uint64_t old = wg_cas((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)((key << PLO_GVB) | 1ull));
            if (old == v) return 1u;
            continue;
        }
        s = (s + 1u) & mask;
    }
    return 0u;
}
__device__ __forceinline__ uint32_t gtab_find(const uint64_t *tab, uint64_t key, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < (1u << 22); ++pr) {
        uint64_t v = gload64(&tab[s]);
        if ((v >> PLO_GVB) == key) return (uint32_t)(v & PLO_GVMASK);
        if (v == PLO_GEMPTY) return 0u;
        s = (s + 1u) & mask;
    }
    return 0u;
}
// value add with claim of EMPTY slots only (ProgramGen multiset; no dead slots there)
__device__ __forceinline__ bool gtab_add(uint64_t *tab, uint64_t key, uint32_t incv, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < PLO_GPROBE_INS; ++pr) {
        uint64_t v = gload64(&tab[s]);
        if (v == PLO_GEMPTY) {
            uint64_t old = wg_cas((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)((key << PLO_GVB) | incv));
            if (old == v) return true;
            continue;
        }
        if ((v >> PLO_GVB) == key) { wg_add((unsigned long long *)&tab[s], (unsigned long long)incv); return true; }
        s = (s + 1u) & mask;
    }
    return false;
}

// set a flag bit on the key's value (insert the key if absent); idempotent, unlike an add
__device__ __forceinline__ bool gtab_flag(uint64_t *tab, uint64_t key, uint32_t flag, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < PLO_GPROBE_INS; ++pr) {
        uint64_t v = gload64(&tab[s]);
        if (v == PLO_GEMPTY) {
            uint64_t old = wg_cas((unsigned long long *)&tab[s], (unsigned long long)v, (unsigned long long)((key << PLO_GVB) | flag));
            if (old == v) return true;
            continue;
        }
        if ((v >> PLO_GVB) == key) { wg_or((unsigned long long *)&tab[s], (unsigned long long)flag); return true; }
        s = (s + 1u) & mask;
    }
    return false;
}

// frequency[key] -= d; returns the frequency before (0 = key not found)
__device__ __forceinline__ uint32_t gtab_subn(uint64_t *tab, uint64_t key, uint32_t d, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < (1u << 22); ++pr) {
        uint64_t v = gload64(&tab[s]);
        if ((v >> PLO_GVB) == key) { uint64_t old = wg_add((unsigned long long *)&tab[s], (unsigned long long)(0ull - (uint64_t)d)); return (uint32_t)(old & PLO_GVMASK); }
        if (v == PLO_GEMPTY) return 0u;
        s = (s + 1u) & mask;
    }
    return 0u;
}
// A probe round looks at PLO_GWIN consecutive slots at once (their loads are in flight together).  Measured on config 5:
// 1 slot per round 648-659 candidates/s, 2 slots 634 -- the extra load instructions cost more than the rounds they save
// (the table phases run at the memory system's request rate).
#ifndef PLO_GWIN
#define PLO_GWIN 1u
#endif
#ifndef PLO_FLU
#define PLO_FLU 2u       /* aggregated entries per thread and trip of the flush */
#endif
// frequency[key] += d (claims the first empty or dead slot in probe order); returns the frequency before, 0xFFFFFFFF = table full
__device__ __forceinline__ uint32_t gtab_addn(uint64_t *tab, uint64_t key, uint32_t d, uint32_t hbits) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s = ghash(key, hbits);
    for (uint32_t pr = 0; pr < PLO_GPROBE_INS; pr += PLO_GWIN) {
        uint64_t v[PLO_GWIN];
#pragma unroll
        for (uint32_t j = 0; j < PLO_GWIN; ++j) v[j] = gload64(&tab[(s + j) & mask]);
        uint32_t adv = PLO_GWIN; bool live = true;
#pragma unroll
        for (uint32_t j = 0; j < PLO_GWIN; ++j) {
            if (!live) continue;
            const uint32_t t = (s + j) & mask;
            if ((v[j] >> PLO_GVB) == key) { uint64_t old = wg_add((unsigned long long *)&tab[t], (unsigned long long)d); return (uint32_t)(old & PLO_GVMASK); }
            if ((v[j] & PLO_GVMASK) == 0ull) {
                uint64_t old = wg_cas((unsigned long long *)&tab[t], (unsigned long long)v[j], (unsigned long long)((key << PLO_GVB) | d));
                if (old == v[j]) return 0u;
                adv = j; live = false;                           // somebody took it: look again from this slot
            }
        }
        s = (s + adv) & mask;
    }
    return 0xFFFFFFFFu;
}

// N insertions of one thread in lock step (second flush pass: PLO_FLU entries per trip; the table is far larger than the
// caches, every probe is a memory round trip, and only independent ones overlap).  live[q] = false: key q is skipped.
template <int N> __device__ __forceinline__ void gtab_addnN(uint64_t *tab, const uint64_t (&key)[N], const uint32_t (&d)[N], const bool (&live)[N], uint32_t hbits, uint32_t (&o)[N]) {
    const uint32_t mask = (1u << hbits) - 1u;
    uint32_t s[N]; bool pend[N];
#pragma unroll
    for (int q = 0; q < N; ++q) { s[q] = ghash(key[q], hbits); pend[q] = live[q]; o[q] = 0xFFFFFFFFu; }
    for (uint32_t pr = 0; pr < PLO_GPROBE_INS; pr += PLO_GWIN) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < N; ++q) any |= pend[q];
        if (!any) break;
        uint64_t v[N][PLO_GWIN];
#pragma unroll
        for (int q = 0; q < N; ++q)
#pragma unroll
            for (uint32_t j = 0; j < PLO_GWIN; ++j) v[q][j] = pend[q] ? gload64(&tab[(s[q] + j) & mask]) : 0ull;
        // where every key goes: its own slot (add), the first empty or dead slot of the window (claim), or on
        uint32_t t[N], at[N]; uint64_t ex[N], res[N];                     // at: 0 move on, 1 add, 2 claim
#pragma unroll
        for (int q = 0; q < N; ++q) {
            at[q] = 0; t[q] = 0; ex[q] = 0; res[q] = 0;
            if (pend[q]) {
#pragma unroll
                for (uint32_t j = 0; j < PLO_GWIN; ++j) if (at[q] == 0u) {
                    if ((v[q][j] >> PLO_GVB) == key[q]) { at[q] = 1u; t[q] = (s[q] + j) & mask; }
                    else if ((v[q][j] & PLO_GVMASK) == 0ull) { at[q] = 2u; t[q] = (s[q] + j) & mask; ex[q] = v[q][j]; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            if (at[q] == 1u) res[q] = wg_add((unsigned long long *)&tab[t[q]], (unsigned long long)d[q]);
            else if (at[q] == 2u) res[q] = wg_cas((unsigned long long *)&tab[t[q]], (unsigned long long)ex[q], (unsigned long long)((key[q] << PLO_GVB) | d[q]));
        }
#pragma unroll
        for (int q = 0; q < N; ++q) if (pend[q]) {
            if (at[q] == 1u) { o[q] = (uint32_t)(res[q] & PLO_GVMASK); pend[q] = false; }
            else if (at[q] == 2u) { if (res[q] == ex[q]) { o[q] = 0u; pend[q] = false; } else s[q] = t[q]; }   // somebody took it: look again from this slot
            else s[q] = (s[q] + PLO_GWIN) & mask;
        }
    }
}

// LDS aggregation table: the rows rewritten by one CSE step retire / create the same triples many times
// (x21 on config 5); the duplicates are summed here and each distinct triple costs one global atomic.
// key48<<16 | count16, open addressing, at most 16 probes; `false` = no room, the caller goes to HBM directly.
#define PLO_AGG_PROBES 16u
#ifdef PLO_BIG_PROFILE
__device__ unsigned long long g_prof2[40];     // phase clocks (100 MHz ticks) by step size class [4][8], summed over candidates; [32] candidates
__device__ unsigned long long g_prof[16];      // thread 0 of every workgroup, sweep 1: cycles per stage (racy sums; profile only)
#define PROF_T(k_) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t__ = clock64(); if (tid == 0) g_prof[k_] += t__ - tp; tp = t__; } while (0)
#else
#define PROF_T(k_) do { } while (0)
#endif
#define PLO_AGG_LIST (PLO_BIG_SELCAP * 4u)     // slot list (u16) kept in the tie-selection buffer, idle during the sweeps
__device__ __forceinline__ bool agg_add(uint64_t *agg, uint32_t aggbits, uint32_t acb, uint64_t key, uint32_t hk, uint32_t *aggn, uint16_t *agglist, uint32_t listcap) {
    const uint32_t mask = (1u << aggbits) - 1u;
    const uint64_t EMPTY = ~0ull << acb;
    uint32_t s = (hk * 0x9E3779B1u) >> (32u - aggbits);       // hk: 32 bits that determine the key ((column, ratio) on config 5)
    uint32_t claimed = 0xFFFFFFFFu; bool done = false;
    // two slots per trip (both LDS reads in flight together): the wave pays the longest probe sequence of its lanes
    for (uint32_t pr = 0; pr < PLO_AGG_PROBES;) {
        const uint32_t s1 = (s + 1u) & mask;
        const uint64_t v0 = __hip_atomic_load(&agg[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), v1 = __hip_atomic_load(&agg[s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_read_b64 (a volatile read would be a flat load)
        const bool hit0 = (v0 >> acb) == key, emp0 = v0 == EMPTY, first = hit0 || emp0;
        const bool hit = hit0 || (!emp0 && (v1 >> acb) == key), emp = emp0 || (!hit && v1 == EMPTY);
        const uint32_t t = first ? s : s1;
        if (hit) { wg_add((unsigned long long *)&agg[t], 1ull); done = true; break; }
        if (emp) {
            const uint64_t old = wg_cas((unsigned long long *)&agg[t], (unsigned long long)EMPTY, (unsigned long long)((key << acb) | 1ull));
            if (old == EMPTY) { claimed = t; done = true; break; }
            continue;                      // somebody took the slot: look at both again
        }
        s = (s + 2u) & mask; pr += 2u;
    }
    // new entries: remember their slots (the flush walks the entries, not the table); one counter update per wave
    const unsigned long long cm = __builtin_amdgcn_ballot_w64(claimed != 0xFFFFFFFFu);
    if (cm) {
        const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(cm);
        uint32_t base = 0;
        if (lane == leader) base = wg_add(aggn, (uint32_t)__builtin_popcountll(cm));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (claimed != 0xFFFFFFFFu) {
            const uint32_t idx = base + (uint32_t)__builtin_popcountll(cm & ((1ull << lane) - 1ull));
            if (idx < listcap) agglist[idx] = (uint16_t)claimed;
        }
    }
    return done;
}

// Mode 2 (at most 32 distinct values, hence at most 1024 distinct ratios v_i/v_j): an aggregation entry is
// (column << 10 | ratio identifier) in a u32 key array and a u16 count array -- 6 bytes per entry instead of 8, 32-bit
// LDS operations, and no modular product in the sweep (the identifier comes from a 2-byte table lookup).
#ifdef PLO_BIG_DIRECT
#define PLO_DIRECT_ONLY(x_) x_
#else
#define PLO_DIRECT_ONLY(x_)
#endif
#define PLO_BIG_QUEUE 128u                  // entries of a wave's queue of the flat sweep ("direct counts")
#define PLO_BIG_QUEUE_WORDS (PLO_BIG_QUEUE * (PLO_BIG_THREADS / 64u))
#ifndef PLO_BIG_DIRECT_MIN
#define PLO_BIG_DIRECT_MIN 128u            // steps of fewer rows use the hashed table only (their flush does not scan the direct table)
#endif
#define PLO_RIDB 10u
#define PLO_RSTRIDE 32u                    // row stride of the ratio-identifier table (at most 32 values): an index is a shift and an or
#ifdef PLO_BIG_DIRECT
struct BigTabs { const uint2 *vts; const uint16_t *rtid; const uint32_t *rval; const uint16_t *invid; uint16_t *list; uint32_t *bloom; const uint8_t *negidx; };
#else
struct BigTabs { const uint2 *vts; const uint16_t *rtid; const uint32_t *rval; const uint16_t *invid; uint16_t *list; uint32_t *bloom; };
#endif   // bloom: DEFER, followed by the scratch region   // list: mode 2, one u16 per aggregation slot
__device__ __forceinline__ bool agg_add_rid(uint32_t *aggk, uint32_t *aggc32, uint32_t aggbits, uint32_t key, uint32_t *aggn, uint16_t *agglist, uint32_t listcap) {
    const uint32_t mask = (1u << aggbits) - 1u;
    uint32_t s = (key * 0x9E3779B1u) >> (32u - aggbits);
    uint32_t claimed = 0xFFFFFFFFu; bool done = false;
    for (uint32_t pr = 0; pr < PLO_AGG_PROBES;) {
        const uint32_t s1 = (s + 1u) & mask;
        const uint32_t k0 = __hip_atomic_load(&aggk[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), k1 = __hip_atomic_load(&aggk[s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool hit0 = k0 == key, emp0 = k0 == 0xFFFFFFFFu, first = hit0 || emp0;
        const bool hit = hit0 || (!emp0 && k1 == key), emp = emp0 || (!hit && k1 == 0xFFFFFFFFu);
        const uint32_t t = first ? s : s1;
        if (hit) { wg_add(&aggc32[t >> 1], 1u << ((t & 1u) << 4)); done = true; break; }
        if (emp) {
            const uint32_t old = wg_cas(&aggk[t], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { if (old != key) claimed = t; wg_add(&aggc32[t >> 1], 1u << ((t & 1u) << 4)); done = true; break; }
            continue;                      // somebody took the slot for another key: look at both again
        }
        s = (s + 2u) & mask; pr += 2u;
    }
    const unsigned long long cm = __builtin_amdgcn_ballot_w64(claimed != 0xFFFFFFFFu);
    if (cm) {
        const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(cm);
        uint32_t base = 0;
        if (lane == leader) base = wg_add(aggn, (uint32_t)__builtin_popcountll(cm));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (claimed != 0xFFFFFFFFu) {
            const uint32_t idx = base + (uint32_t)__builtin_popcountll(cm & ((1ull << lane) - 1ull));
            if (idx < listcap) agglist[idx] = (uint16_t)claimed;
        }
    }
    return done;
}

// The same probe loop for the staged sweep (mode 2 with deferred updates): a claimed slot sets its bit in a bitmap with a
// fire-and-forget atomic -- no counter with a returned value, no slot list (two LDS round trips less on the claim path).
__device__ __forceinline__ bool agg_add_rid_bm(uint32_t *aggk, uint32_t *aggc32, uint32_t aggbits, uint32_t key, uint32_t *bm, uint32_t *iters = nullptr) {
    // aligned pairs of slots (one ds_read_b64, the two counts share a word); the hash is a 24-bit product (full rate)
    const uint32_t mask = (1u << aggbits) - 1u;
    uint32_t s = ((uint32_t)__umul24(key, 0x9E3779u) >> (32u - aggbits)) & ~1u;      // (__umul24 returns int: an arithmetic shift without the cast)
    for (uint32_t pr = 0; pr < PLO_AGG_PROBES;) {
        if (iters) ++*iters;
        const unsigned long long kk = __hip_atomic_load((unsigned long long *)(aggk + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t k0 = (uint32_t)kk, k1 = (uint32_t)(kk >> 32);
        if (k0 == key || k1 == key) { wg_add(&aggc32[s >> 1], k0 == key ? 1u : 0x10000u); return true; }      // (an empty first slot rules the second one out)
        if (k0 == 0xFFFFFFFFu || k1 == 0xFFFFFFFFu) {
            const uint32_t sec = k0 == 0xFFFFFFFFu ? 0u : 1u, t = s + sec;
            const uint32_t old = wg_cas(&aggk[t], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { if (old != key) wg_or(&bm[t >> 5], 1u << (t & 31u)); wg_add(&aggc32[s >> 1], sec ? 0x10000u : 1u); return true; }
            continue;                      // somebody took the slot for another key: look at both again
        }
        s = (s + 2u) & mask; pr += 2u;
    }
    return false;
}

// First round of the same probe, written without a loop (round 4).  One trip of the flat sweep finds 64 keys; 99.8 % of them are in
// their home pair or claim a slot of it, so the common case is ONE pair read and at most three predicated LDS operations: the add on
// a hit, the compare-and-swap of an empty slot, the bitmap bit of a fresh claim.  The probe loop above -- breaks, a `continue` after a
// lost claim, a result flag -- compiled to ~60 scalar and ~70 vector instructions of exec-mask bookkeeping per trip, which is what the
// sweep was bound by (4 waves per SIMD issue one instruction per cycle group each: 108 VALU + 100 SALU per trip).  Returns true when
// the key is NOT settled (both slots hold other keys, or the claim lost to another key): the caller sends those lanes, ~0.2 % of them,
// through the loop.
__device__ __forceinline__ bool agg_add_rid_first(uint32_t *aggk, uint32_t *aggc32, uint32_t aggbits, uint32_t key, uint32_t *bm) {
    const uint32_t s = ((uint32_t)__umul24(key, 0x9E3779u) >> (32u - aggbits)) & ~1u;
    const unsigned long long kk = __hip_atomic_load((unsigned long long *)(aggk + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t k0 = (uint32_t)kk, k1 = (uint32_t)(kk >> 32);
    const bool hit0 = k0 == key, hit = hit0 || k1 == key, emp0 = k0 == 0xFFFFFFFFu, tryc = !hit && (emp0 || k1 == 0xFFFFFFFFu);
    const uint32_t sec = hit ? (hit0 ? 0u : 1u) : (emp0 ? 0u : 1u), t = s + sec;
    uint32_t old = 0u;
    if (tryc) old = wg_cas(&aggk[t], 0xFFFFFFFFu, key);
    const bool fresh = tryc && old == 0xFFFFFFFFu, ok = hit || fresh || (tryc && old == key);      // (old == key: a lane with the same key claimed the slot first)
    if (fresh) wg_or(&bm[t >> 5], 1u << (t & 31u));
    if (ok) wg_add(&aggc32[s >> 1], sec ? 0x10000u : 1u);
    return !ok;
}

#ifdef PLO_BIG_PROFILE
// the same with a clock after every LDS round trip (profile build): acc[0] pair read, [1] compare-and-swap, [2] bitmap + count
__device__ __forceinline__ bool agg_add_rid_first_prof(uint32_t *aggk, uint32_t *aggc32, uint32_t aggbits, uint32_t key, uint32_t *bm, unsigned long long *acc) {
    const unsigned long long c0 = clock64();
    const uint32_t s = ((uint32_t)__umul24(key, 0x9E3779u) >> (32u - aggbits)) & ~1u;
    const unsigned long long kk = __hip_atomic_load((unsigned long long *)(aggk + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t k0 = (uint32_t)kk, k1 = (uint32_t)(kk >> 32);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long c1 = clock64();
    const bool hit0 = k0 == key, hit = hit0 || k1 == key, emp0 = k0 == 0xFFFFFFFFu, tryc = !hit && (emp0 || k1 == 0xFFFFFFFFu);
    const uint32_t sec = hit ? (hit0 ? 0u : 1u) : (emp0 ? 0u : 1u), t = s + sec;
    uint32_t old = 0u;
    if (tryc) old = wg_cas(&aggk[t], 0xFFFFFFFFu, key);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long c2 = clock64();
    const bool fresh = tryc && old == 0xFFFFFFFFu, ok = hit || fresh || (tryc && old == key);
    if (fresh) wg_or(&bm[t >> 5], 1u << (t & 31u));
    if (ok) wg_add(&aggc32[s >> 1], sec ? 0x10000u : 1u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long c3 = clock64();
    acc[0] += c1 - c0; acc[1] += c2 - c1; acc[2] += c3 - c2;
    return !ok;
}
#endif

// packed row entry: column (15 bits) | +-1 flag (bit 15) | value index (16 bits)
#define PLO_ECOL(e_) ((e_) & 0x7FFFu)
#define PLO_EUNIT(e_) (((e_) >> 15) & 1u)
#define PLO_EVI(e_) ((e_) >> 16)
// position of column c in row [base, base+L) (sorted by column), or -1
__device__ __forceinline__ int row_find(const uint32_t *ent, uint32_t base, uint32_t L, uint32_t c) {
    uint32_t lo = 0, hi = L;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (PLO_ECOL(ent[base + mid]) < c) lo = mid + 1; else hi = mid; }
    return (lo < L && PLO_ECOL(ent[base + lo]) == c) ? (int)lo : -1;
}

struct BigShared {
    uint32_t M, theta, ncols, nbadd, nbmul, nmult, naff, dmcount, hlcount, rng, errflag, sel_n, sel_over, invr, fullscans, rebuilds, steps, hlbad, acc0, acc1;
    uint32_t a, b, r, aggn, nspill, keepn; uint64_t kprime; uint64_t selkey;
#ifdef PLO_BIG_DIRECT
    uint32_t dn;                   // words of the direct count table the step's sweep has touched (listed for the flush)
#endif
    uint32_t nbisect, spilltot, listover, nwin, nsearched;   // nwin: windows of the flat sweep beyond the first of a batch of rows (thread 0's wave); nsearched: rows walked by the row search   // diagnostics: tie picks by bisection, entries through the spill list, sweeps whose slot list overflowed
    uint32_t logn, hotn, hotbits, nforced, hotops, logtot_lo, logtot_hi;   // DEFER: log fill, claimed hot slots, hot table size; diagnostics: merges forced by log/hot pressure, updates served by the hot table, log entries written
    uint32_t derr; unsigned long long tmg[4], tmb[4]; uint32_t ngrp; uint32_t outcnt[64];           // DEFER merge: live entries written back per partition of the current group
    uint32_t cblk[512];            // level-M triple counts per block of 64 first columns (NCmax <= 32768)
    unsigned long long tph[8];     // phase clocks (100 MHz ticks): level, select, rows, sweep1, flush1, sweep2, flush2, tail
#ifdef PLO_BIG_PROFILE
    unsigned long long tb1[4], tb2[4]; uint32_t nb[4], fb1, fb2, fl1, fl2;   // sweep clocks by step size class, fallbacks, flushed keys
    unsigned long long tpc[4][8];  // phase clocks by step size class
    unsigned long long pw[16];      // sweep of the big steps, summed over waves: cycles waiting for the chunk + stores, aggregation, loop overhead; trips; wave time; waves
#endif
    uint32_t part[8];
    uint64_t sel[PLO_BIG_SELCAP];
};

#define BSYNC() __syncthreads()
#ifdef PLO_BIG_NOFENCE
#define PLO_SCHED_FENCE() do { } while (0)
#else
#define PLO_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)      /* the machine scheduler keeps the stages of the hand-ordered trip apart */
#endif
#ifdef PLO_BIG_PROFILE
#define PLO_STAMP(q_) do { if (threadIdx.x == 0) { unsigned long long t_ = wall_clock64(); sh.tph[q_] += t_ - tstamp; sh.tpc[sh.M >= 256u ? 0 : sh.M >= 64u ? 1 : sh.M >= 16u ? 2 : 3][q_] += t_ - tstamp; tstamp = t_; } } while (0)
#else
#define PLO_STAMP(q_) do { if (threadIdx.x == 0) { unsigned long long t_ = wall_clock64(); sh.tph[q_] += t_ - tstamp; tstamp = t_; } } while (0)
#endif

// ===========================================================================
// Deferred cold updates (template parameter DEFER).  Measured on config 5 (tests/micro/defer_model.cpp): of the 1.2e7
// retirements and 3.6e6 insertions of a candidate, 99.5 % touch triples whose frequency is below the current window of
// levels [theta, M] -- only ~8 k triples are ever "hot" at a time -- yet every one of them cost a random 64-byte line of
// a 64 MB table, both ways: 1.9 of the 3.1 GB of HBM traffic per candidate.  Decisions (max level, tie set, counts at
// level M) need exact frequencies only inside the window.  So:
//   hot     open-addressing table of the triples with frequency >= theta (and of spilled insertions), exact, small
//           (2^16 slots); a Bloom filter in LDS (2^17 bits) says "certainly cold" for almost every other key.
//   log     a retirement / insertion of a cold triple is ONE 8-byte record key48 | insert flag | d15 appended to a
//           sequential log (wave-coalesced stores): no read, no random access.
//   store   the cold triples, 2^pbits partitions by hash prefix, each an unordered array (key48<<16 | count16).
//   merge   when M falls below theta (or the log / hot table fill up): hot entries go to the log; the log is split by
//           partition through an LDS staging buffer (runs of ~64 bytes); every partition (group of partitions when they
//           are small) is summed in an LDS hash table (counts biased by 2^15: records arrive in any order), live
//           triples (frequency >= 2) are written back compactly and counted per level; the new window [theta', M] is
//           chosen on the exact histogram and its triples move to the hot table.  All of it is streaming traffic.
// Frequencies never rise after the step that creates a triple, so a cold triple stays cold until the next merge and the
// hot table always holds every triple of frequency >= theta: levels >= theta of hist[] stay exact, lower levels are
// not maintained between merges (never read: a scan that would go below theta triggers the merge).
// Same results as the eager table, bit for bit (tests/test_gpu_l32cut.py, test_gpu_config5.py, test_gpu_cse_hbm.py).
// ===========================================================================
#define PLO_DLB 13u                 /* LDS summing table of a merge: 2^13 slots of 8 bytes */
#define PLO_DCH 4096u               /* log records per round of the partition pass */
#define PLO_DBIAS 0x8000u
#define PLO_DBLOOM_WORDS 4096u      /* 2^17 bits */
#define PLO_DMREG_WORDS 16384u      /* Bloom filter + scratch region = the merge's 64 KB of LDS */
#define PLO_DPMAX 2048u             /* partitions: counters, offsets and tails of the partition pass live behind the staging buffer */
#define PLO_LEMPTY (~0ull)
static_assert(PLO_DPMAX <= 4u * PLO_BIG_THREADS && PLO_DCH % PLO_BIG_THREADS == 0u && 2u * PLO_DCH + 3u * PLO_DPMAX <= PLO_DMREG_WORDS, "merge layout");

__device__ __forceinline__ uint64_t dmix(uint64_t key) { return key * 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ uint32_t dpart(uint64_t key, uint32_t pbits) { return pbits ? (uint32_t)(dmix(key) >> (64u - pbits)) : 0u; }
__device__ __forceinline__ uint32_t dlslot(uint64_t key, uint32_t pbits, uint32_t lb) { return (uint32_t)((dmix(key) << pbits) >> (64u - lb)); }
__device__ __forceinline__ bool dbloom_test(const uint32_t *bl, uint64_t key) {
    const uint64_t g = key * 0xD6E8FEB86659FD93ull;
    const uint32_t i1 = (uint32_t)(g >> 47), i2 = (uint32_t)(g >> 30) & 0x1FFFFu, i3 = (uint32_t)(g >> 13) & 0x1FFFFu;
    const uint32_t w1 = __hip_atomic_load(&bl[i1 >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), w2 = __hip_atomic_load(&bl[i2 >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                   w3 = __hip_atomic_load(&bl[i3 >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (((w1 >> (i1 & 31u)) & (w2 >> (i2 & 31u)) & (w3 >> (i3 & 31u))) & 1u) != 0u;
}
__device__ __forceinline__ void dbloom_set(uint32_t *bl, uint64_t key) {
    const uint64_t g = key * 0xD6E8FEB86659FD93ull;
    const uint32_t i1 = (uint32_t)(g >> 47), i2 = (uint32_t)(g >> 30) & 0x1FFFFu, i3 = (uint32_t)(g >> 13) & 0x1FFFFu;
    wg_or(&bl[i1 >> 5], 1u << (i1 & 31u)); wg_or(&bl[i2 >> 5], 1u << (i2 & 31u)); wg_or(&bl[i3 >> 5], 1u << (i3 & 31u));
}
// slot of `key` in the hot table (v = its word), 0xFFFFFFFF when absent.  No slot is ever emptied between merges.
__device__ __forceinline__ uint32_t hot_slot(const uint64_t *hot, uint64_t key, uint32_t hb, uint64_t &v) {
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t s = ghash(key, hb);
    for (uint32_t pr = 0; pr <= mask; ++pr) {
        v = gload64(&hot[s]);
        if ((v >> PLO_GVB) == key) return s;
        if (v == PLO_GEMPTY) return 0xFFFFFFFFu;
        s = (s + 1u) & mask;
    }
    return 0xFFFFFFFFu;
}
// frequency[key] += d in the hot table, claiming EMPTY slots only; returns the frequency before (0 = new key), 0xFFFFFFFF = table full
__device__ __forceinline__ uint32_t hot_addn(uint64_t *hot, uint64_t key, uint32_t d, uint32_t hb, uint32_t *hotn) {
    const uint32_t mask = (1u << hb) - 1u;
    uint32_t s = ghash(key, hb);
    for (uint32_t pr = 0; pr <= 2u * mask + 1u; ++pr) {
        const uint64_t v = gload64(&hot[s]);
        if ((v >> PLO_GVB) == key) { const uint64_t old = wg_add((unsigned long long *)&hot[s], (unsigned long long)d); return (uint32_t)(old & PLO_GVMASK); }
        if (v == PLO_GEMPTY) {
            const uint64_t old = wg_cas((unsigned long long *)&hot[s], (unsigned long long)v, (unsigned long long)((key << PLO_GVB) | d));
            if (old == v) { wg_add(hotn, 1u); return 0u; }
            continue;                                   // somebody took it: look at the slot again
        }
        s = (s + 1u) & mask;
    }
    return 0xFFFFFFFFu;
}
// log records: key48 << 16 | insert flag (bit 15) | d (15 bits)
#define PLO_DREC(key_, d_, ins_) (((uint64_t)(key_) << 16) | ((ins_) ? 0x8000ull : 0ull) | (uint64_t)((d_) & 0x7FFFu))
// wave-collective append of up to three records per lane (all lanes of the wave call it): one counter update per wave,
// the records of each kind are stored side by side
__device__ __forceinline__ void dlog_append3(uint64_t *dlog, uint32_t *logn, uint32_t logcap, bool h1, uint64_t e1, bool h2, uint64_t e2, bool h3, uint64_t e3, uint32_t *errflag) {
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(h1), m2 = __builtin_amdgcn_ballot_w64(h2), m3 = __builtin_amdgcn_ballot_w64(h3);
    const uint32_t n1 = (uint32_t)__builtin_popcountll(m1), n2 = (uint32_t)__builtin_popcountll(m2), n3 = (uint32_t)__builtin_popcountll(m3);
    if (n1 + n2 + n3 == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t base = 0;
    if (lane == 0u) base = wg_add(logn, n1 + n2 + n3);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base + n1 + n2 + n3 > logcap) { if (lane == 0u) wg_max(errflag, (uint32_t)BERR_TABLE); return; }
    if (h1) dlog[base + (uint32_t)__builtin_popcountll(m1 & below)] = e1;
    if (h2) dlog[base + n1 + (uint32_t)__builtin_popcountll(m2 & below)] = e2;
    if (h3) dlog[base + n1 + n2 + (uint32_t)__builtin_popcountll(m3 & below)] = e3;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, o); if ((int)lane >= o) v += t; }
    return v;
}
// add `delta` to the biased count of `key` in the LDS summing table (find or claim)
__device__ __forceinline__ bool lt_put(uint64_t *ltab, uint64_t key, int32_t delta, uint32_t pbits, uint32_t lb) {
    const uint32_t mask = (1u << lb) - 1u;
    uint32_t s = dlslot(key, pbits, lb);
    for (uint32_t pr = 0; pr <= 2u * mask + 1u; ++pr) {
        const uint64_t v = __hip_atomic_load(&ltab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((v >> 16) == key) { wg_add((unsigned long long *)&ltab[s], (unsigned long long)(long long)delta); return true; }
        if (v == PLO_LEMPTY) {
            const uint64_t old = wg_cas((unsigned long long *)&ltab[s], (unsigned long long)PLO_LEMPTY, (unsigned long long)((key << 16) | (uint64_t)(uint32_t)((int32_t)PLO_DBIAS + delta)));
            if (old == PLO_LEMPTY) return true;
            continue;
        }
        s = (s + 1u) & mask;
    }
    return false;
}

// The merge (all threads of the workgroup).  `mreg` = the Bloom filter followed by the scratch region (PLO_DMREG_WORDS words).
// first = true: the store is the plan's image and hist[] its histogram, only the window is chosen.
__device__ __forceinline__ void defer_merge(const BigPlan &P, uint8_t *ws, BigShared &sh, uint32_t *hist, uint32_t *mreg, bool first)
{
    const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63u, wave = tid >> 6, nwaves = nth >> 6;
    uint64_t *store = (uint64_t *)(ws + P.o_store), *dlog = (uint64_t *)(ws + P.o_log), *hot = (uint64_t *)(ws + P.o_hot), *HL = (uint64_t *)(ws + P.o_hl);
    uint32_t *pcount = (uint32_t *)(ws + P.o_pcount), *ptail = (uint32_t *)(ws + P.o_ptail);
    const uint32_t pbits = P.pbits, Pn = 1u << pbits, capp = P.capp, rcap = P.capp + P.plcap;
    unsigned long long tm0 = wall_clock64();
#define PLO_MSTAMP(q_) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); sh.tmg[q_] += t_ - tm0; tm0 = t_; } } while (0)
    if (!first) {
        // ---- 0. the hot triples go back through the log (they are absent from the store)
        {
            const uint32_t hslots = 1u << sh.hotbits;
            for (uint32_t s0 = 0; s0 < hslots; s0 += nth) {
                const uint32_t s = s0 + tid;
                const uint64_t v = s < hslots ? gload64(&hot[s]) : PLO_GEMPTY;
                const bool have = v != PLO_GEMPTY && (uint32_t)(v & PLO_GVMASK) >= 2u;
                dlog_append3(dlog, &sh.logn, P.logcap, have, (v & ~PLO_GVMASK) | 0x8000ull | (v & 0x7FFFull), false, 0ull, false, 0ull, &sh.errflag);
            }
        }
        PLO_BIG_FENCE(); BSYNC();
        PLO_MSTAMP(0);
        // ---- A. partition pass: PLO_DCH records per round are ranked per partition (LDS counters), laid out by partition in the
        // staging buffer and written behind the partitions' logs: neighbouring lanes store neighbouring words
        {
            uint64_t *stage = (uint64_t *)mreg; uint32_t *cnt = mreg + 2u * PLO_DCH, *pos = cnt + PLO_DPMAX, *tail = pos + PLO_DPMAX;
            for (uint32_t q = tid; q < Pn; q += nth) tail[q] = pcount[q];
            const uint32_t nlog = sh.logn < P.logcap ? sh.logn : P.logcap;
            constexpr uint32_t U = PLO_DCH / PLO_BIG_THREADS;
            const uint32_t K = (Pn + nth - 1u) / nth;                       // partitions per thread in the prefix sum (<= 4)
            for (uint32_t base = 0; base < nlog; base += PLO_DCH) {
                const uint32_t n = nlog - base < PLO_DCH ? nlog - base : PLO_DCH;
                for (uint32_t q = tid; q < Pn; q += nth) cnt[q] = 0u;
                BSYNC();
                uint64_t e[U]; uint32_t pp[U], rk[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) { const uint32_t idx = u * nth + tid; e[u] = idx < n ? dlog[base + idx] : 0ull; }
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) { const uint32_t idx = u * nth + tid; pp[u] = 0u; rk[u] = 0u; if (idx < n) { pp[u] = dpart(e[u] >> 16, pbits); rk[u] = wg_add(&cnt[pp[u]], 1u); } }
                BSYNC();
                {   // exclusive prefix sum of the counters
                    uint32_t loc[4] = {0u, 0u, 0u, 0u}, sum = 0u;
                    for (uint32_t k = 0; k < K; ++k) { const uint32_t q = tid * K + k; const uint32_t v = q < Pn ? cnt[q] : 0u; loc[k & 3u] = sum; sum += v; }
                    const uint32_t inc = wave_incl_scan(sum);
                    if (lane == 63u) sh.part[wave] = inc;
                    BSYNC();
                    uint32_t woff = 0u;
                    for (uint32_t w = 0; w < wave; ++w) woff += sh.part[w];
                    for (uint32_t k = 0; k < K; ++k) { const uint32_t q = tid * K + k; if (q < Pn) pos[q] = woff + inc - sum + loc[k & 3u]; }
                }
                BSYNC();
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) { const uint32_t idx = u * nth + tid; if (idx < n) stage[pos[pp[u]] + rk[u]] = e[u]; }
                for (uint32_t q = tid; q < Pn; q += nth) {
                    const uint32_t c = cnt[q], t = tail[q];
                    if (t + c > rcap) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 101u); }
                    tail[q] = t + c; cnt[q] = t - pos[q];                  // position in the partition's log = cnt[q] + position in the staging buffer
                }
                BSYNC();
                for (uint32_t j = tid; j < n; j += nth) {
                    const uint64_t ee = stage[j];
                    const uint32_t q = dpart(ee >> 16, pbits), o = cnt[q] + j;
                    if (o < rcap) store[(uint64_t)q * rcap + o] = ee;
                }
                BSYNC();
            }
            for (uint32_t q = tid; q < Pn; q += nth) ptail[q] = tail[q] < rcap ? tail[q] : rcap;
        }
        PLO_BIG_FENCE(); BSYNC();
        PLO_MSTAMP(1);
        if (sh.errflag) return;
        // ---- B. every group of partitions is summed in LDS; live triples go back to the store, the histogram is recounted
        {
            for (uint32_t f = tid; f <= P.maxf0; f += nth) hist[f] = 0u;
            uint64_t *ltab = (uint64_t *)mreg;
            BSYNC();
            // Counts of 128 partitions ahead live in two registers (lane i: partition W + i, W + 64 + i); the bounds of a group come from
            // them without a memory round trip.  The first four records of a thread for the NEXT group are requested while the table of
            // the current group is scanned and written back.
            uint32_t W = 0, cW = lane < Pn ? ptail[lane] : 0u, cW2 = 64u + lane < Pn ? ptail[64u + lane] : 0u;
            auto bounds = [&](uint32_t p_, uint32_t &cT, uint32_t &g, uint32_t &tot) {
                const uint32_t o = p_ - W + lane;                                  // 0 .. 127
                const uint32_t x0 = (uint32_t)__shfl((int)cW, (int)(o & 63u)), x1 = (uint32_t)__shfl((int)cW2, (int)(o & 63u));
                cT = o < 64u ? x0 : x1;
                const uint32_t inc = wave_incl_scan(cT);
                const unsigned long long okm = __builtin_amdgcn_ballot_w64(inc <= P.lgrp);
                g = okm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okm);
                if (g == 0u) g = 1u;
                if (g > Pn - p_) g = Pn - p_;
                tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(g - 1u));
            };
            uint32_t p = 0, cT = 0, g = 0, tot = 0;
            bounds(0u, cT, g, tot);
            // records tid, tid + nth, tid + 2 nth, tid + 3 nth of the group's FIRST partition (the others, and anything beyond, are loaded in the loop)
            uint64_t f0 = 0, f1 = 0, f2 = 0, f3 = 0;
            auto first4 = [&](uint32_t p_, uint32_t n0) {
                const uint64_t *sp = store + (uint64_t)p_ * rcap;
                f0 = tid < n0 ? sp[tid] : 0ull; f1 = tid + nth < n0 ? sp[tid + nth] : 0ull; f2 = tid + 2u * nth < n0 ? sp[tid + 2u * nth] : 0ull; f3 = tid + 3u * nth < n0 ? sp[tid + 3u * nth] : 0ull;
            };
            first4(0u, (uint32_t)__builtin_amdgcn_readlane((int)cT, 0));
#define PLO_PUT(v_) do { const int32_t d_ = (int32_t)((v_) & 0x7FFFull); if (d_) if (!lt_put(ltab, (v_) >> 16, ((v_) & 0x8000ull) ? d_ : -d_, pbits, lb)) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 103u); } } while (0)
            unsigned long long tbl = wall_clock64();
            while (p < Pn) {
                if (tot > (7u << (PLO_DLB - 3u))) { if (tid == 0) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 102u); } break; }       // more than 7/8 of the table
                uint32_t lb = 6u; while ((1u << lb) < 2u * tot + 64u && lb < PLO_DLB) ++lb;
                for (uint32_t s = tid; s < (1u << lb); s += nth) ltab[s] = PLO_LEMPTY;
                if (tid < 64u) sh.outcnt[tid] = 0u;
                BSYNC();
                unsigned long long tb0 = wall_clock64();
#define PLO_BSTAMP(q_) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); sh.tmb[q_] += t_ - tb0; tb0 = t_; } } while (0)
                if (tid == 0) { sh.tmb[3] += tb0 - tbl; ++sh.ngrp; }
                PLO_PUT(f0); PLO_PUT(f1); PLO_PUT(f2); PLO_PUT(f3);
                for (uint32_t j = 0; j < g; ++j) {
                    const uint32_t nT = (uint32_t)__builtin_amdgcn_readlane((int)cT, (int)j);
                    const uint64_t *sp = store + (uint64_t)(p + j) * rcap;
                    for (uint32_t e = tid + (j == 0u ? 4u * nth : 0u); e < nT; e += 4u * nth) {
                        const uint64_t v0 = sp[e], v1 = e + nth < nT ? sp[e + nth] : 0ull, v2 = e + 2u * nth < nT ? sp[e + 2u * nth] : 0ull, v3 = e + 3u * nth < nT ? sp[e + 3u * nth] : 0ull;
                        PLO_PUT(v0); PLO_PUT(v1); PLO_PUT(v2); PLO_PUT(v3);
                    }
                }
                // the next group: slide the window of counts, bounds, first records
                const uint32_t pn = p + g;
                if (pn >= W + 64u) { W += 64u; cW = cW2; const uint32_t q = W + 64u + lane; cW2 = q < Pn ? ptail[q] : 0u; }
                uint32_t cTn = 0, gn = 1, totn = 0;
                if (pn < Pn) bounds(pn, cTn, gn, totn);
                BSYNC();
                PLO_BSTAMP(0);
                if (pn < Pn) first4(pn, (uint32_t)__builtin_amdgcn_readlane((int)cTn, 0)); else { f0 = f1 = f2 = f3 = 0ull; }
                for (uint32_t s = tid; s < (1u << lb); s += nth) {
                    const uint64_t v = ltab[s];
                    if (v == PLO_LEMPTY) continue;
                    const uint32_t cb = (uint32_t)(v & 0xFFFFull);
                    if (cb < PLO_DBIAS + 2u) continue;                       // frequency below 2: never chosen, never rises -- dropped
                    const uint32_t c = cb - PLO_DBIAS; const uint64_t k = v >> 16;
                    if (c > P.maxf0) { wg_max(&sh.errflag, (uint32_t)BERR_FREQ); continue; }
                    wg_add(&hist[c], 1u);
                    const uint32_t j = dpart(k, pbits) - p;
                    const uint32_t idx = j < 64u ? wg_add(&sh.outcnt[j], 1u) : capp;
                    if (idx < capp) store[(uint64_t)(p + j) * rcap + idx] = (k << 16) | 0x8000ull | c; else { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 104u); }
                }
                BSYNC();
                PLO_BSTAMP(1);
                if (tid < g) { const uint32_t c_ = sh.outcnt[tid] < capp ? sh.outcnt[tid] : capp; pcount[p + tid] = c_; ptail[p + tid] = c_; }
                tbl = wall_clock64();
                p = pn; cT = cTn; g = gn; tot = totn;
            }
#undef PLO_PUT
        }
        PLO_BIG_FENCE(); BSYNC();
        PLO_MSTAMP(2);
        if (sh.errflag) return;
    }
    // ---- the new window [theta, M]: at most hwin triples (the whole top level in any case), then the hot table of its triples
    if (tid == 0) {
        uint32_t Mx = P.M0; while (Mx >= 2u && hist[Mx] == 0u) --Mx;
        uint64_t acc = 0; uint32_t th = 2u;
        if (Mx >= 2u) { th = Mx; for (uint32_t f = Mx; f >= 2u; --f) { if (f != Mx && acc + hist[f] > P.hwin) break; acc += hist[f]; th = f; } }
        if (acc > P.hlcap / 2u) wg_max(&sh.errflag, (uint32_t)BERR_HL);
        uint32_t hb = P.hotbits_min; while ((1ull << hb) < 4ull * acc + 1024ull && hb < P.hotbits_max) ++hb;
#ifdef PLO_BIG_MERGELOG
        printf("# merge %u at step %u: M %u theta %u window %llu triples (hot table 2^%u); histogram of the levels below: [%u]=%u [%u]=%u [%u]=%u; log %u records\n", sh.fullscans, sh.steps, Mx, th, (unsigned long long)acc, hb,
               th > 2u ? th - 1u : 0u, th > 2u ? hist[th - 1u] : 0u, th > 3u ? th - 2u : 0u, th > 3u ? hist[th - 2u] : 0u, 2u, hist[2], sh.logn);
#endif
        sh.M = Mx >= 2u ? Mx : 0u; sh.theta = th; sh.hotbits = hb; sh.hotn = 0u; sh.hlcount = 0u; sh.logn = 0u; sh.hlbad = 0u; ++sh.fullscans;
    }
    BSYNC();
    if (sh.errflag) return;
    {
        const uint32_t hb = sh.hotbits, th = sh.theta;
        for (uint32_t s = tid; s < (1u << hb); s += nth) hot[s] = PLO_GEMPTY;
        for (uint32_t w = tid; w < PLO_DBLOOM_WORDS; w += nth) mreg[w] = 0u;
        PLO_BIG_FENCE(); BSYNC();
        if (sh.M >= 2u) {
            for (uint32_t q = wave; q < Pn; q += nwaves) {                    // a wave streams whole partitions: no barrier in this pass
                const uint32_t n = pcount[q];
                uint64_t *sp = store + (uint64_t)q * rcap;
                for (uint32_t e0 = 0; e0 < n; e0 += 256u) {
                    uint64_t v[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) { const uint32_t e = e0 + u * 64u + lane; v[u] = e < n ? sp[e] : 0ull; }
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const uint32_t e = e0 + u * 64u + lane, c = (uint32_t)(v[u] & 0x7FFFull);
                        if (e < n && c >= th) {
                            const uint64_t k = v[u] >> 16;
                            if (hot_addn(hot, k, c, hb, &sh.hotn) != 0u) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 106u); }
                            dbloom_set(mreg, k);
                            const uint32_t idx = wg_add(&sh.hlcount, 1u);
                            if (idx < P.hlcap) HL[idx] = k; else wg_max(&sh.errflag, (uint32_t)BERR_HL);
                            sp[e] = v[u] & ~0x7FFFull;                        // left the store (a flagged record of frequency 0)
                        }
                    }
                }
            }
        }
    }
    PLO_BIG_FENCE(); BSYNC();
    PLO_MSTAMP(3);
#undef PLO_MSTAMP
}

// ---------------------------------------------------------------------------
// One candidate by one workgroup.  Returns (adds<<32 | muls) in thread 0.
// ---------------------------------------------------------------------------
template <int MODE, bool DEFER, bool IDK> __device__ __forceinline__ uint64_t big_candidate(const BigPlan &P, uint8_t *ws, uint64_t seed, BigShared &sh, uint32_t *hist, uint64_t *agg, uint32_t aggbits, const BigTabs &TB, uint32_t *errw)
{
    const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63u, wave = tid >> 6, nwaves = nth >> 6;
    uint64_t *tab   = (uint64_t *)(ws + (DEFER ? P.o_hot : P.o_tab));      // the table the level code and the tie pick look triples up in (DEFER: the hot table)
    uint64_t *dlog  = (uint64_t *)(ws + P.o_log); uint32_t *bloom = TB.bloom;   // DEFER
    uint32_t *ent   = (uint32_t *)(ws + P.o_ent);
    uint32_t *len   = (uint32_t *)(ws + P.o_len), *ucount = (uint32_t *)(ws + P.o_ucount), *cntM = (uint32_t *)(ws + P.o_cntM);
    uint64_t *DM    = (uint64_t *)(ws + P.o_dm), *HL = (uint64_t *)(ws + P.o_hl);
    // {value, inverse} per value index: LDS copy, or global memory above 512 values.  Two typed accesses,
    // never one generic pointer: a flat load waits for vmcnt(0) AND lgkmcnt(0) and would drain every prefetch.
    // (MODE is a template parameter: the compiler turns a run-time choice between the two address spaces into a flat load)
    const uint2 *vtg = P.vt; const uint2 *vts = TB.vts;
    auto VT = [&](uint32_t vi) -> uint2 { if constexpr (MODE == 1) return vts[vi]; else return vtg[vi]; };
    const uint16_t *rtid = TB.rtid, *invid = TB.invid; const uint32_t *rval = TB.rval;   // mode 2
    uint32_t *aggk = (uint32_t *)agg, *aggc32 = aggk + (1u << aggbits); uint16_t *aggc16 = (uint16_t *)aggc32;      // mode 2: key array, count array
    // Direct counts (round 4, mode 2 with deferred updates): behind the hashed table, one word per column c < P.dcols -- low half: entries
    // of ratio 1 met by the running sweep, high half: ratio -1 -- and a queue of PLO_BIG_QUEUE words per wave (see the flat sweep)
#ifdef PLO_BIG_DIRECT
    uint32_t *dcnt = aggc32 + (1u << aggbits) / 2u, *wqueue = dcnt + P.dcols;
#endif
    uint32_t *aff   = (uint32_t *)(ws + P.o_aff), *ncrptr = (uint32_t *)(ws + P.o_ncrptr), *ncr = (uint32_t *)(ws + P.o_ncr);
    // slots claimed in the aggregation table by the running sweep (the flush walks this list, not the table): mode 2 has room
    // for every slot; the other modes keep a short list in the tie-selection buffer, idle during the sweeps, and walk the
    // table when a step claims more
    constexpr bool FAST = MODE == 2 && DEFER;                                  // staged aggregation with deferred claims, bitmap of claimed slots
    uint32_t *aggbm = (uint32_t *)sh.sel;                                       // FAST: one bit per aggregation slot (2^aggbits <= 2^14 bits; the tie-selection buffer is idle during the sweeps)
    uint16_t *agglist = (MODE == 2 && !DEFER) ? TB.list : (uint16_t *)sh.sel; const uint32_t listcap = (MODE == 2 && !DEFER) ? (1u << aggbits) : PLO_AGG_LIST;   // (DEFER: the Bloom filter has the place of mode 2's full list)
    uint64_t *spill = (uint64_t *)(ws + P.o_spill); const uint32_t spillcap = P.nnz + 64u;   // new-column pairs of entries that found no room in LDS (a step touches every entry at most once)
    uint32_t *multc = (uint32_t *)(ws + P.o_multc), *multv = (uint32_t *)(ws + P.o_multv);
    // A pair key is (first column, second column, ratio) in 48 bits.  The ratio field holds the residue (rb bits) -- or, IDK (mode 2 only:
    // a modulus too wide for 48 bits, e.g. a 31-bit prime with 32768 columns), the ratio's IDENTIFIER (kb <= 10 bits).  Identifiers are
    // ranks in the sorted list of ratios, so keys compare as they do with residues (the tie pick walks triples in key order, :244-253).
    const uint32_t p = P.p, rb = P.rb, kb = IDK ? P.kb : P.rb, abits = kb + P.bb, n = P.n, m = P.m, mers = P.mers;
    uint32_t hbits = DEFER ? P.hotbits_min : P.hbits;                                // DEFER: the hot table is sized anew at every merge
    const uint64_t mu = P.mu, cap = 1ull << P.hbits;
    auto key_ratio = [&](uint32_t x) -> uint32_t {                                   // the ratio field of a key for the residue x
        if constexpr (MODE == 2 && IDK) { uint32_t lo = 0, hi = P.nr; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rval[mid] < x) lo = mid + 1u; else hi = mid; } return lo; }
        else return x;
    };
#define BKEY(a_, b_, r_) (((uint64_t)(a_) << abits) | ((uint64_t)(b_) << kb) | (uint64_t)key_ratio(r_))

    // ---- load the candidate image
    {   // 16-byte copies, 4 in flight per thread (all buffers are 256-byte aligned, sizes padded by the host)
        const uint4 *s4 = (const uint4 *)P.tab0; uint4 *d4 = (uint4 *)tab;
        const uint64_t n4 = DEFER ? 0ull : cap >> 1;                                                // (DEFER: the store image is copied partition by partition below)
        for (uint64_t s = tid; s < n4; s += 4ull * nth) {
            uint4 x0 = s4[s], x1, x2, x3; const bool b1 = s + nth < n4, b2 = s + 2ull * nth < n4, b3 = s + 3ull * nth < n4;
            if (b1) x1 = s4[s + nth]; if (b2) x2 = s4[s + 2ull * nth]; if (b3) x3 = s4[s + 3ull * nth];
            d4[s] = x0; if (b1) d4[s + nth] = x1; if (b2) d4[s + 2ull * nth] = x2; if (b3) d4[s + 3ull * nth] = x3;
        }
        const uint32_t q4 = (P.nnz + 3u) >> 2;
        const uint4 *e4 = (const uint4 *)P.ent0; uint4 *de = (uint4 *)ent;
        for (uint32_t k = tid; k < q4; k += 2u * nth) { uint4 x = e4[k], y; const bool b1 = k + nth < q4; if (b1) y = e4[k + nth]; de[k] = x; if (b1) de[k + nth] = y; }
    }
    if constexpr (DEFER) {
        uint32_t *pcount = (uint32_t *)(ws + P.o_pcount), *ptail = (uint32_t *)(ws + P.o_ptail);
        for (uint32_t q = tid; q < (1u << P.pbits); q += nth) { pcount[q] = P.pcount0[q]; ptail[q] = P.pcount0[q]; }
        uint64_t *store = (uint64_t *)(ws + P.o_store); const uint32_t rcap = P.capp + P.plcap;
        for (uint32_t q = wave; q < (1u << P.pbits); q += nwaves) {           // a wave copies whole partitions of the image (stride capp) to the store (stride capp + plcap)
            const uint32_t nq = P.pcount0[q]; const uint64_t *sp = P.st0 + (uint64_t)q * P.capp; uint64_t *dp = store + (uint64_t)q * rcap;
            for (uint32_t e = lane; e < nq; e += 128u) { const uint64_t x0 = sp[e], x1 = e + 64u < nq ? sp[e + 64u] : 0ull; dp[e] = x0; if (e + 64u < nq) dp[e + 64u] = x1; }
        }
    }
    uint32_t *tl = (uint32_t *)(ws + P.o_tl), *clen = (uint32_t *)(ws + P.o_clen), *keep = (uint32_t *)(ws + P.o_keep);
    {   // this candidate's copy of the input columns' row lists (compacted in place as the steps walk them)
        const uint32_t q4 = (P.nnz + 3u) >> 2;
        const uint4 *t4 = (const uint4 *)P.trows; uint4 *d4 = (uint4 *)tl;
        for (uint32_t k = tid; k < q4; k += 2u * nth) { uint4 x = t4[k], y; const bool b1 = k + nth < q4; if (b1) y = t4[k + nth]; d4[k] = x; if (b1) d4[k + nth] = y; }
    }
    for (uint32_t i = tid; i < m; i += nth) len[i] = P.rs[i + 1] - P.rs[i];
    for (uint32_t c = tid; c < P.NCmax; c += nth) { ucount[c] = c < n ? P.ucount0[c] : 0u; cntM[c] = 0u; clen[c] = c < n ? P.tptr[c + 1] - P.tptr[c] : 0u; }
    for (uint32_t f = tid; f <= P.maxf0; f += nth) hist[f] = P.hist0[f];
    const uint32_t acb = P.agg_cb; const uint64_t AEMPTY = ~0ull << acb;
    if constexpr (MODE == 2) { for (uint32_t s = tid; s < (1u << aggbits); s += nth) { aggk[s] = 0xFFFFFFFFu; aggc16[s] = 0; } }
    else for (uint32_t s = tid; s < (1u << aggbits); s += nth) agg[s] = AEMPTY;
#ifdef PLO_BIG_DIRECT
    if constexpr (MODE == 2 && DEFER) for (uint32_t s = tid; s < P.dcols; s += nth) dcnt[s] = 0u;
#endif
    if (tid == 0) {
        uint64_t x = seed + 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        sh.rng = 1u + (uint32_t)(x % 2147483646ull);
        sh.M = P.M0; sh.theta = P.M0 + 1u; sh.ncols = n; sh.nbadd = 0; sh.nbmul = 0; sh.nmult = 0; sh.dmcount = 0; sh.hlcount = 0;
        for (int q = 0; q < 8; ++q) sh.tph[q] = 0;
#ifdef PLO_BIG_PROFILE
        for (int q = 0; q < 4; ++q) { sh.tb1[q] = sh.tb2[q] = 0; sh.nb[q] = 0; } sh.fb1 = sh.fb2 = sh.fl1 = sh.fl2 = 0; for (int q = 0; q < 16; ++q) sh.pw[q] = 0; for (int c_ = 0; c_ < 4; ++c_) for (int q = 0; q < 8; ++q) sh.tpc[c_][q] = 0;
#endif
        sh.errflag = 0; sh.fullscans = 0; sh.rebuilds = 0; sh.steps = 0; sh.hlbad = 0; ncrptr[0] = 0; sh.nbisect = 0; sh.spilltot = 0; sh.listover = 0; sh.nwin = 0; sh.nsearched = 0;
        sh.logn = 0; sh.hotn = 0; sh.hotbits = P.hotbits_min; sh.derr = 0; for (int q = 0; q < 4; ++q) { sh.tmg[q] = 0; sh.tmb[q] = 0; } sh.ngrp = 0; sh.nforced = 0; sh.hotops = 0; sh.logtot_lo = 0; sh.logtot_hi = 0;
    }
    PLO_BIG_FENCE(); BSYNC();
    bool dfirst = true;                                                       // DEFER: the first merge only chooses the window

    bool need_rebuild = true;
    unsigned long long tstamp = wall_clock64();
    for (;;) {
        if (sh.errflag) break;
        // ---- level bookkeeping: lower M while empty (thread 0), decide on a window rescan
        if constexpr (DEFER) {
            // levels >= theta are exact; a scan that would go below theta, a full log or a full hot table call for a merge
            for (;;) {
                if (tid == 0) {
                    uint32_t M = sh.M; const uint32_t th = sh.theta;
                    while (M >= th && hist[M] == 0u) --M;
                    const bool low = M < th, pressure = sh.logn > P.logtrig || sh.hotn > (1u << sh.hotbits) / 2u || sh.hlbad;
                    sh.part[0] = (M != sh.M) ? 1u : 0u; sh.part[1] = (low || pressure) ? 1u : 0u;
                    if (!low) { sh.M = M; if (pressure) ++sh.nforced; }
                }
                BSYNC();
                const bool mg = sh.part[1] != 0u;
                if (sh.part[0]) need_rebuild = true;
                BSYNC();
                if (!mg) break;
                { const uint32_t ln_ = sh.logn; if (tid == 0) { const uint32_t lo_ = sh.logtot_lo; sh.logtot_lo = lo_ + ln_; if (sh.logtot_lo < lo_) ++sh.logtot_hi; } }
                defer_merge(P, ws, sh, hist, bloom, dfirst);
                if (!dfirst) {                                                // the merge worked in the scratch region: the aggregation table is empty again
                    if constexpr (MODE == 2) { for (uint32_t s = tid; s < (1u << aggbits); s += nth) { aggk[s] = 0xFFFFFFFFu; aggc16[s] = 0; } }
                    else for (uint32_t s = tid; s < (1u << aggbits); s += nth) agg[s] = AEMPTY;
#ifdef PLO_BIG_DIRECT
                    if constexpr (MODE == 2 && DEFER) for (uint32_t s = tid; s < P.dcols; s += nth) dcnt[s] = 0u;
#endif
                    BSYNC();
                }
                dfirst = false; need_rebuild = true; hbits = sh.hotbits;
                if (sh.errflag || sh.M <= 1u) break;
            }
            if (sh.errflag) break;
        } else {
        if (tid == 0) {
            uint32_t M = sh.M;
            while (M >= 2u && hist[M] == 0u) --M;
            sh.part[0] = (M != sh.M) ? 1u : 0u;
            sh.M = M;
        }
        BSYNC();
        }
        const uint32_t M = sh.M;
        if (M <= 1u) break;                                               // OneSub :255
        if (!DEFER && sh.part[0]) need_rebuild = true;
        BSYNC();
        if (need_rebuild) {
            if (!DEFER && (M < sh.theta || sh.hlbad)) {
                // full table scan: new window [theta', M] holding at most hlcap/2 keys
                if (tid == 0) {
                    uint64_t acc = 0; uint32_t th = M;
                    for (uint32_t f = M; f >= 2u; --f) { if (acc + hist[f] > P.hlcap / 2u) break; acc += hist[f]; th = f; }
                    if (hist[M] > P.hlcap / 2u) wg_max(&sh.errflag, (uint32_t)BERR_HL);
                    sh.theta = th; sh.hlcount = 0; sh.hlbad = 0; ++sh.fullscans;
                }
                BSYNC();
                const uint32_t th = sh.theta;
                for (uint64_t s = tid; s < cap; s += 4ull * nth) {             // 4 independent loads in flight per thread
                    uint64_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (s + (uint64_t)u * nth < cap) ? gload64(&tab[s + (uint64_t)u * nth]) : PLO_GEMPTY;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if ((uint32_t)(v[u] & PLO_GVMASK) >= th && v[u] != PLO_GEMPTY) {
                            uint32_t idx = wg_add(&sh.hlcount, 1u);
                            if (idx < P.hlcap) HL[idx] = v[u] >> PLO_GVB; else wg_max(&sh.errflag, (uint32_t)BERR_HL);
                        }
                }
                BSYNC();
            }
            // per-column counters and key list of the current level from the window list; keys that fell below
            // the window are dropped from it (they can never come back)
            for (uint32_t c = tid; c < sh.ncols; c += nth) cntM[c] = 0u;
            for (uint32_t c = tid; c < 512u; c += nth) sh.cblk[c] = 0u;
            const uint32_t hn = sh.hlcount < P.hlcap ? sh.hlcount : P.hlcap;
            PLO_BIG_FENCE(); BSYNC();
            if (tid == 0) { sh.dmcount = 0; sh.hlcount = 0; ++sh.rebuilds; }
            BSYNC();
            uint64_t *HL2 = HL + P.hlcap;                                       // ping-pong halves of the window list
            const uint32_t th2 = sh.theta;
            for (uint32_t k = tid; k < hn; k += nth) {
                const uint64_t key = HL[k];
                const uint32_t c = gtab_find(tab, key, hbits);
                if (c >= th2) HL2[wg_add(&sh.hlcount, 1u)] = key;
                if (c == M) {
                    const uint32_t fc = (uint32_t)(key >> abits);
                    wg_add(&cntM[fc], 1u); wg_add(&sh.cblk[fc >> 6], 1u);
                    uint32_t idx = wg_add(&sh.dmcount, 1u);
                    if (idx < P.dmcap) DM[idx] = key; else wg_max(&sh.errflag, (uint32_t)BERR_DM);
                }
            }
            PLO_BIG_FENCE(); BSYNC();
            { const uint32_t hn2 = sh.hlcount; for (uint32_t k = tid; k < hn2; k += nth) HL[k] = HL2[k]; }
            need_rebuild = false;
            PLO_BIG_FENCE(); BSYNC();
            if (sh.errflag) break;
        }
        PLO_STAMP(0);
        // ---- tie pick (OneSub :244-265): k-th triple of frequency M in map order
        const uint32_t ncols = sh.ncols;
        {
            if (wave == 0) {
                // block of 64 first columns from the LDS block sums, then one wave-wide load of that block.  The 512 block sums are
                // scanned by the wave (8 per lane, prefix sum over the lanes), not by one thread reading them one after the other.
                uint32_t blk = 0; uint64_t k = 0, run = 0;
                if (lane == 0) {
                    const uint32_t T = hist[M];
                    if (T > 1u) { uint64_t x = 950706376ull * (uint64_t)sh.rng; sh.rng = (uint32_t)(x % 2147483647ull); k = sh.rng % T; }
                    sh.sel_n = 0; sh.sel_over = 0;
                }
                {
                    const uint32_t k32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 0);         // k < T < 2^32
                    uint32_t q8[8], mine = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) { q8[u] = sh.cblk[lane * 8u + (uint32_t)u]; mine += q8[u]; }
                    uint32_t inc = mine;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += t; }
                    const uint64_t hm0 = __ballot(k32 < inc);
                    if (!hm0) { if (lane == 0) wg_max(&sh.errflag, (uint32_t)BERR_SEL); }
                    const uint32_t L0 = hm0 ? (uint32_t)__builtin_ctzll(hm0) : 0u;
                    uint32_t b_ = lane * 8u, r_ = inc - mine;                          // the owner lane walks its 8 sums
                    { bool done = false;
#pragma unroll
                      for (int u = 0; u < 8; ++u) if (!done) { if (k32 < r_ + q8[u]) done = true; else { r_ += q8[u]; ++b_; } } }
                    blk = (uint32_t)__builtin_amdgcn_readlane((int)b_, (int)L0);
                    run = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r_, (int)L0);
                    k = (uint64_t)k32;
                }
                const uint32_t c = blk * 64u + lane;
                uint32_t q = c < ncols ? gload32(&cntM[c]) : 0u, incl = q;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += t; }
                const uint64_t kk = k - run;                                           // (both wave-uniform)
                const uint64_t hm = __ballot(kk < (uint64_t)incl);
                if (!hm) { if (lane == 0) wg_max(&sh.errflag, (uint32_t)BERR_SEL); }
                else if (lane == (uint32_t)__builtin_ctzll(hm)) { sh.a = c; sh.kprime = kk - (uint64_t)(incl - q); }
            }
            BSYNC();
            if (sh.errflag) break;
            const uint32_t a = sh.a;
            const uint32_t dn = sh.dmcount < P.dmcap ? sh.dmcount : P.dmcap;
            for (uint32_t k = tid; k < dn; k += nth) {
                const uint64_t key = DM[k];
                if ((uint32_t)(key >> abits) == a && gtab_find(tab, key, hbits) == M) {
                    uint32_t idx = wg_add(&sh.sel_n, 1u);
                    if (idx < P.selcap) sh.sel[idx] = key; else sh.sel_over = 1u;
                }
            }
            BSYNC();
            if (sh.sel_over) {
                // rare: more ties in one column than the LDS list holds -> bisection on the key value
                if (tid == 0) ++sh.nbisect;
                uint64_t lo = (uint64_t)a << abits, hi = (((uint64_t)a + 1ull) << abits) - 1ull;
                while (lo < hi) {
                    const uint64_t mid = lo + ((hi - lo) >> 1);
                    if (tid == 0) sh.sel_n = 0;
                    BSYNC();
                    uint32_t c = 0;
                    for (uint32_t k = tid; k < dn; k += nth) {
                        const uint64_t key = DM[k];
                        if ((uint32_t)(key >> abits) == a && key <= mid && gtab_find(tab, key, hbits) == M) ++c;
                    }
                    if (c) wg_add(&sh.sel_n, c);
                    BSYNC();
                    const uint32_t tot = sh.sel_n;
                    BSYNC();
                    if ((uint64_t)tot >= sh.kprime + 1ull) hi = mid; else lo = mid + 1ull;
                }
                if (tid == 0) sh.selkey = lo;
            } else {
                const uint32_t sn = sh.sel_n;
                for (uint32_t x = tid; x < sn; x += nth) {
                    const uint64_t mine = sh.sel[x]; uint32_t rank = 0;
                    for (uint32_t y = 0; y < sn; ++y) rank += (sh.sel[y] < mine) ? 1u : 0u;
                    if ((uint64_t)rank == sh.kprime) sh.selkey = mine;
                }
                if (tid == 0 && (uint64_t)sn <= sh.kprime) wg_max(&sh.errflag, (uint32_t)BERR_SEL);
            }
            BSYNC();
            if (sh.errflag) break;
        }
        const uint64_t key = sh.selkey;
        PLO_STAMP(1);
        const uint32_t rfield = (uint32_t)(key & ((1ull << kb) - 1ull)), b = (uint32_t)(key >> kb) & ((1u << P.bb) - 1u), a = (uint32_t)(key >> abits);
        uint32_t r = rfield; if constexpr (MODE == 2 && IDK) r = rval[rfield];
        const uint32_t lm = ncols;
        if (lm + 1u >= P.NCmax) { if (tid == 0) wg_max(&sh.errflag, (uint32_t)BERR_COLS); BSYNC(); break; }
        // ---- RemOneCSE :60-194
        const bool swap = gload32(&ucount[a]) < gload32(&ucount[b]);      // :70-88
        const uint32_t l0 = swap ? b : a, l1 = swap ? a : b;
        if (tid == 0) { sh.naff = 0; sh.aggn = 0; sh.nspill = 0; sh.keepn = 0; PLO_DIRECT_ONLY(sh.dn = 0;) }      // (nothing here uses `swap`: the two counts stay in flight while the row lists are walked)
        if constexpr (FAST) { for (uint32_t w = tid; w < ((1u << aggbits) + 31u) / 32u; w += nth) aggbm[w] = 0u; }      // (the tie pick used the buffer)
        BSYNC();
#define RL(v_, k_) ((uint32_t)__builtin_amdgcn_readlane((int)(v_), (int)(k_)))
        {   // rows holding the triple: walk the shorter row list of the two columns
            uint32_t *la = a < n ? tl + P.tptr[a] : ncr + ncrptr[a - n], *lb = b < n ? tl + P.tptr[b] : ncr + ncrptr[b - n];
            const uint32_t na = gload32(&clen[a]), nb = gload32(&clen[b]);          // live lengths: what earlier walks left of the lists
            const bool walk_a = na <= nb;
            const uint32_t *lst = walk_a ? la : lb; const uint32_t ln = walk_a ? na : nb;
            if (tid == 0) sh.nsearched += ln;
            // Two rows per thread and trip, both columns of both rows searched in lock step: the four binary searches have
            // their loads in flight together (8 dependent memory round trips for two rows instead of 36).  A search keeps
            // the last entry it saw at its upper bound: when it ends that is the entry at the found position.
            uint32_t *newrows = ncr + ncrptr[lm - n];
            auto emit = [&](uint32_t i, uint32_t base, uint32_t L, uint32_t pa, uint32_t pb, uint32_t ea, uint32_t eb) -> bool {      // true: the row holds the triple
                uint32_t inv_r;
                if constexpr (MODE == 2) {
                    if (rval[rtid[PLO_EVI(eb) * PLO_RSTRIDE + PLO_EVI(ea)]] != r) return false;
                    inv_r = rval[rtid[PLO_EVI(ea) * PLO_RSTRIDE + PLO_EVI(eb)]];
                } else {
                    const uint2 A = VT(PLO_EVI(ea)), B = VT(PLO_EVI(eb));
                    if (B.x != bmul(r, A.x, p, mu, mers)) return false;
                    inv_r = bmul(A.x, B.y, p, mu, mers);
                }
                const uint32_t idx = wg_add(&sh.naff, 1u);
#ifdef PLO_BIG_PROFILE
                atomicAdd(&sh.tb2[1], 100ull * L);
#endif
                if constexpr (MODE == 2) {
                    // 16-byte record (value indices have 5 bits): positions, row start, length (<= 8192: 14 bits) | value index and +-1 flag of
                    // the a and b entries, row -- 32 bytes in round 2: 48 MB less written and read per candidate on config 5
                    // (bits 26-31: the value index of -v_a and "there is one" -- the direct counts of the flat sweep)
#ifdef PLO_BIG_DIRECT
                    const uint32_t nva = TB.negidx[PLO_EVI(ea)];
#else
                    const uint32_t nva = 0xFFu;
#endif
                    *(uint4 *)(aff + 4u * idx) = make_uint4(pa | (pb << 16), base, L | (PLO_EUNIT(ea) << 14) | (PLO_EVI(ea) << 15) | (PLO_EUNIT(eb) << 20) | (PLO_EVI(eb) << 21) | ((nva & 31u) << 26) | (nva != 0xFFu ? 0x80000000u : 0u), i);
                } else {
                uint32_t *rec = aff + 8u * idx;                             // record: row, positions (16 bits each), row start and length; the two packed entries
                *(uint4 *)rec = make_uint4(i, pa | (pb << 16), base, L);
                *(uint2 *)(rec + 4) = make_uint2(ea, eb);
                }
                newrows[idx] = i;                                              // row list of the new column (4-byte entries: the records are 32 bytes apart)
                if (idx == 0) sh.invr = inv_r;                                   // 1/r
                len[i] = L - 1u;                                               // the sweep works from the record
                if (PLO_EUNIT(ea)) wg_sub(&ucount[a], 1u);                     // :70-77 counts, kept incrementally
                if (PLO_EUNIT(eb)) wg_sub(&ucount[b], 1u);
                if (PLO_EUNIT(swap ? eb : ea)) wg_add(&ucount[lm], 1u);
                return true;
            };
            // a searched row stays in the walked list iff it still holds the list's column and does not lose it in this step; the kept
            // rows go to a scratch list and are copied to the front of the walked one behind the barrier below (no order is needed)
#ifdef PLO_BIG_NOCOMPACT
            auto keep_row = [&](uint32_t, bool, bool) { };                      // (A/B switch: the lists keep their stale rows, as in rounds 1-3)
#else
            auto keep_row = [&](uint32_t i, bool has_col, bool affected) { if (has_col && !affected) keep[wg_add(&sh.keepn, 1u)] = i; };
#endif
#ifdef PLO_BIG_PROFILE
            if (tid == 0) { sh.tb2[0] += 100ull * ln; sh.tb2[2] += 100ull * (na <= nb ? nb : na); }
#endif
#ifdef PLO_BIG_QUATSEARCH
            // (round 4 experiment, measured SLOWER: row search 91 -> 141 ms per candidate at full load, 49 -> 66 ms alone -- the phase is bound by the number of
            // scattered 4-byte requests a CU can issue, not by the number of dependent rounds: 6 loads x 4 rounds cost more than 2 x 7.)
            // One row per thread, both columns searched in lock step by QUATERNARY search: three probes per search and round, all
            // six loads in flight together, a quarter of the range left -- 3 dependent memory round trips for a row of 64 entries
            // (5 for 1024) where the binary search of rounds 1-3 needed 7 (11).  The phase is latency-bound (a step's rows are
            // searched once, by idle threads mostly), so the extra requests are free.  A search keeps the entry it saw at its
            // upper bound: when it ends that is the entry at the found position.
            for (uint32_t k = tid; k < ln; k += nth) {
                const uint32_t i0 = lst[k];
                const uint32_t bs0 = P.rs[i0], L0 = len[i0];
                uint32_t lo[2] = {0u, 0u}, hi[2] = {L0, L0}, ev[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
                for (;;) {
                    uint32_t v[2][3], pq[2][3]; bool any = false;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t w = hi[q] - lo[q]; const bool go = w != 0u; any |= go;
#pragma unroll
                        for (int j = 0; j < 3; ++j) { pq[q][j] = lo[q] + (((uint32_t)(j + 1) * w) >> 2); v[q][j] = go ? ent[bs0 + pq[q][j]] : 0u; }
                    }
                    if (!any) break;
#pragma unroll
                    for (int q = 0; q < 2; ++q) if (hi[q] != lo[q]) {
                        const uint32_t c = q ? b : a;
                        if (PLO_ECOL(v[q][0]) >= c) { hi[q] = pq[q][0]; ev[q] = v[q][0]; }
                        else if (PLO_ECOL(v[q][1]) >= c) { lo[q] = pq[q][0] + 1u; hi[q] = pq[q][1]; ev[q] = v[q][1]; }
                        else if (PLO_ECOL(v[q][2]) >= c) { lo[q] = pq[q][1] + 1u; hi[q] = pq[q][2]; ev[q] = v[q][2]; }
                        else lo[q] = pq[q][2] + 1u;
                    }
                }
                // (an upper bound that never moved is the row length: ev stays all ones, whose column field matches no column)
                {   const bool fa = PLO_ECOL(ev[0]) == a && lo[0] < L0, fb = PLO_ECOL(ev[1]) == b && lo[1] < L0;
                    const bool aff0 = fa && fb && emit(i0, bs0, L0, lo[0], lo[1], ev[0], ev[1]);
                    keep_row(i0, walk_a ? fa : fb, aff0); }
            }
#else
            for (uint32_t k = tid; k < ln; k += 2u * nth) {
                const bool two = k + nth < ln;
                const uint32_t i0 = lst[k], i1 = two ? lst[k + nth] : i0;
                const uint32_t bs0 = P.rs[i0], bs1 = P.rs[i1], L0 = len[i0], L1 = two ? len[i1] : 0u;
                uint32_t lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {L0, L0, L1, L1}, ev[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                for (;;) {
                    uint32_t v[4], mid[4]; bool any = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        mid[q] = (lo[q] + hi[q]) >> 1; const bool go = lo[q] < hi[q]; any |= go;
                        v[q] = go ? ent[(q < 2 ? bs0 : bs1) + mid[q]] : 0u;
                    }
                    if (!any) break;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (lo[q] < hi[q]) { if (PLO_ECOL(v[q]) < ((q & 1) ? b : a)) lo[q] = mid[q] + 1u; else { hi[q] = mid[q]; ev[q] = v[q]; } }
                }
                // (an upper bound that never moved is the row length: ev stays all ones, whose column field matches no column)
                {   const bool fa = PLO_ECOL(ev[0]) == a && lo[0] < L0, fb = PLO_ECOL(ev[1]) == b && lo[1] < L0;
                    const bool aff0 = fa && fb && emit(i0, bs0, L0, lo[0], lo[1], ev[0], ev[1]);
                    keep_row(i0, walk_a ? fa : fb, aff0); }
                if (two) { const bool fa = PLO_ECOL(ev[2]) == a && lo[2] < L1, fb = PLO_ECOL(ev[3]) == b && lo[3] < L1;
                    const bool aff1 = fa && fb && emit(i1, bs1, L1, lo[2], lo[3], ev[2], ev[3]);
                    keep_row(i1, walk_a ? fa : fb, aff1); }
            }
#endif
        }
        PLO_BIG_FENCE(); BSYNC();
        const uint32_t naff = sh.naff;
#ifndef PLO_BIG_NOCOMPACT
        {   // the walked list, compacted (it is read next in a later step, many barriers from here)
            const uint32_t nk = sh.keepn, lc = gload32(&clen[a]) <= gload32(&clen[b]) ? a : b;
            uint32_t *dst = lc < n ? tl + P.tptr[lc] : ncr + ncrptr[lc - n];
            for (uint32_t k = tid; k < nk; k += nth) dst[k] = keep[k];
            BSYNC();                                                           // (every thread has read both lengths)
            if (tid == 0) clen[lc] = nk;
        }
#endif
        PLO_STAMP(2);
        if (naff != M) { if (tid == 0) wg_max(&sh.errflag, (uint32_t)BERR_FREQ); BSYNC(); break; }   // frequency must equal the row count
        // The sweep over the affected rows: rewrite each row (:96-110) and retire its old pairs (:115-118) in one pass over
        // its entries.  In every affected row v_b = r v_a, so the two
        // pairs an entry (c, v) forms with a and with b are functions of (c, x), x = v_a/v (c < a) or v/v_a (c > a): the
        // pair with a has ratio x, the pair with b has ratio r x (c < a), r/x (a < c < b) or x/r (b < c).  One LDS entry
        // per (c, x) therefore carries both retirements; the flush derives the two table keys.
        {
            // via, vib: value indices of the row's two removed entries (they name v_a and v_b)
#ifdef PLO_BIG_PROFILE
            uint32_t probe_iters = 0; unsigned long long pq[3] = {0, 0, 0}, prt = 0;
#endif
            auto retire_entry = [&](uint32_t e, uint32_t via, uint32_t vib, uint2 VA, uint2 VB, bool noagg = false) {      // noagg: the LDS table had no room for it (already tried)
                const uint32_t c = PLO_ECOL(e);
                // x = v_a/v_c (c < a) or v_c/v_a (c > a) names both retired pairs; y = v_a/v_c names the pair with the new column
                // (x itself, or 1/x: kept beside x in the entry when the bits allow, so that the flush needs no inversion)
                uint32_t x, y, q2, ins;                                        // q2: ratio of the pair with b; ins: ratio of the pair with the new column (both only on the fallback path)
                if constexpr (MODE == 2) {
                    const uint32_t vi = PLO_EVI(e);
#ifdef PLO_BIG_PROFILE
                    const unsigned long long cr0 = clock64();
#endif
                    const uint32_t xid = rtid[c < a ? (via * PLO_RSTRIDE) | vi : (vi * PLO_RSTRIDE) | via];      // one lookup, no branch
#ifdef PLO_BIG_PROFILE
                    if constexpr (FAST) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prt += clock64() - cr0;
                        const bool pend = agg_add_rid_first_prof(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, aggbm, pq);
                        if (!__builtin_amdgcn_ballot_w64(pend)) return;
                        if (!pend || agg_add_rid_bm(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, aggbm, &probe_iters)) return;
                    }
#elif !defined(PLO_BIG_FIRSTPROBE)
                    if constexpr (FAST) { if (!noagg) { if (agg_add_rid_bm(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, aggbm)) return; } }
#else
                    if constexpr (FAST) {
                        if (!noagg) {
                        const bool pend = agg_add_rid_first(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, aggbm);
                        if (!__builtin_amdgcn_ballot_w64(pend)) return;                  // (wave-uniform: nine trips in ten end here)
                        if (!pend || agg_add_rid_bm(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, aggbm)) return;
                        }
                    }
#endif
                    else if (agg_add_rid(aggk, aggc32, aggbits, (c << PLO_RIDB) | xid, &sh.aggn, agglist, listcap)) return;
                    const uint32_t bc = rval[rtid[vib * PLO_RSTRIDE + vi]];                                  // v_b / v_c
                    x = rval[xid]; y = rval[rtid[via * PLO_RSTRIDE + vi]];
                    q2 = c < b ? bc : rval[rtid[vi * PLO_RSTRIDE + vib]];
                    ins = l0 == a ? y : bc;
                } else {
                    const uint2 V = VT(PLO_EVI(e));
                    y = bmul(VA.x, V.y, p, mu, mers);
                    x = c < a ? y : bmul(V.x, VA.y, p, mu, mers);
                    const uint64_t cx = ((uint64_t)c << rb) | x;
                    if (agg_add(agg, aggbits, acb, P.agg_dual ? (cx << rb) | y : cx, (uint32_t)cx ^ ((uint32_t)(cx >> 32) * 0x85EBCA6Bu), &sh.aggn, agglist, listcap)) return;
                    q2 = c < b ? bmul(VB.x, V.y, p, mu, mers) : bmul(V.x, VB.y, p, mu, mers);
                    ins = bmul(l0 == a ? VA.x : VB.x, V.y, p, mu, mers);
                }
#ifdef PLO_BIG_PROFILE
                wg_add(&sh.fb1, 1u);
#endif
                // no room in the LDS table (the key is then absent from it for the whole sweep): retire in HBM directly.
                // A triple that is not in the table had frequency 1 (pruned): nothing to do for it.
                const uint64_t k1 = c < a ? BKEY(c, a, x) : BKEY(a, c, x);
                const uint64_t k2 = c < b ? BKEY(c, b, q2) : BKEY(b, c, q2);
                if constexpr (DEFER) {
                    // one instance of each retired pair: the hot table when the triple is there, else a log record
                    for (int w_ = 0; w_ < 2; ++w_) {
                        const uint64_t kk = w_ ? k2 : k1; const uint32_t fc = w_ ? (c < b ? c : b) : (c < a ? c : a);
                        bool done = false;
                        if (dbloom_test(bloom, kk)) {
                            uint64_t v; const uint32_t sl = hot_slot(tab, kk, hbits, v);
                            if (sl != 0xFFFFFFFFu) {
                                const uint32_t o = (uint32_t)(wg_add((unsigned long long *)&tab[sl], ~0ull) & PLO_GVMASK);
                                if (o) { wg_sub(&hist[o], 1u); if (o > 1u) wg_add(&hist[o - 1u], 1u); if (o == M) { wg_sub(&cntM[fc], 1u); wg_sub(&sh.cblk[fc >> 6], 1u); } }
                                else wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
                                done = true;
                            }
                        }
                        if (!done) { const uint32_t li = wg_add(&sh.logn, 1u); if (li < P.logcap) dlog[li] = PLO_DREC(kk, 1u, false); else wg_max(&sh.errflag, (uint32_t)BERR_TABLE); }
                    }
                    const uint32_t idx = wg_add(&sh.nspill, 1u);
                    if (idx < spillcap) spill[idx] = BKEY(c, lm, ins); else wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
                    return;
                }
                const uint32_t o1 = gtab_dec(tab, k1, hbits);
                if (o1) {
                    wg_sub(&hist[o1], 1u); if (o1 > 1u) wg_add(&hist[o1 - 1u], 1u);
                    if (o1 == M) { wg_sub(&cntM[c < a ? c : a], 1u); wg_sub(&sh.cblk[(c < a ? c : a) >> 6], 1u); }
                }
                const uint32_t o2 = gtab_dec(tab, k2, hbits);
                if (o2) {
                    wg_sub(&hist[o2], 1u); if (o2 > 1u) wg_add(&hist[o2 - 1u], 1u);
                    if (o2 == M) { wg_sub(&cntM[c < b ? c : b], 1u); wg_sub(&sh.cblk[(c < b ? c : b) >> 6], 1u); }
                }
                // its pair with the new column is inserted after all retirements (flush, second pass)
                const uint32_t idx = wg_add(&sh.nspill, 1u);
                if (idx < spillcap) spill[idx] = BKEY(c, lm, ins); else wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
            };
            // The sweep proper.  A wave owns the rows wave, wave + nwaves, ... of the step and first loads the records of its
            // next 64 rows, ONE PER LANE (a record = row start, length, the two positions, the two removed entries), reading
            // them back with v_readlane: no trip waits for a record -> row dependency.  Prefetches are unconditional
            // (clamped index, length 0 past the end): no branch around a load.  In-place rewrite: a chunk's stores reach back
            // at most two positions and never forward, the chunks of a row are worked on in order by one wave, and a chunk
            // is loaded before the chunks before it are stored -- so no load sees a store.
            const uint32_t nrw = naff > wave ? (naff - wave + nwaves - 1u) / nwaves : 0u;      // rows of this wave: wave, wave + nwaves, ...
#ifdef PLO_BIG_PROFILE
            unsigned long long pw0 = 0, pw1 = 0, pw2 = 0, ptr = 0, pit = 0, pact = 0, tl_ = clock64(); const unsigned long long ts_ = tl_;
#endif
#ifndef PLO_BIG_ROWSWEEP
            if constexpr (FAST) {
            // Flat sweep (round 3): the entries of the wave's next 64 rows form ONE sequence and every trip takes its next 64
            // entries, whatever rows they belong to -- lanes are full (two rows per trip, a 64-lane chunk each: 53 % on config 5),
            // and what was wave-uniform per row (positions, length, value indices: scalar registers, spilled) is per lane.
            // Row of a lane: the row ends inside a window of 2048 entries are marked in 32 LDS words of the wave, one per
            // trip; a lane's row = rows that ended before the trip + marks at or below its lane (v_mbcnt); the row's record
            // comes from the lane that loaded it (ds_bpermute).
            // Stores of a trip reach back at most two positions and never into the next trip's entries (requested a trip
            // earlier), so no load sees a store.
            uint64_t *fm = sh.sel + 256u + 32u * wave;                  // (the bitmap of claimed slots takes at most the first 2 KB)
            const uint32_t dump = P.nnz + 64u + lane;                   // (the entry array has 128 spare words)
            // Direct counts (round 4).  On config 5 84 % of the entries carry +-7710: four of five swept entries have ratio 1 or -1 to
            // their row's v_a, and every trip paid the ratio lookup, the pair read and (96 % of the trips: some lane claims) the
            // compare-and-swap for all 64 lanes.  Now an entry of ratio +-1 in a column below P.dcols is ONE fire-and-forget add to the
            // column's word of a direct table (no lookup: v == v_a, or v == -v_a with the index of -v_a carried by the row record);
            // the other entries are appended to the wave's queue (column, value indices: 30 bits) and go through the hashed table 64 at
            // a time, full lanes.  The flush reads both tables.  Steps of fewer than PLO_BIG_DIRECT_MIN rows keep to the hashed table
            // (their flush does not scan the direct one).
#ifdef PLO_BIG_DIRECT
            const bool use_direct = P.dcols != 0u && naff >= PLO_BIG_DIRECT_MIN;
            uint32_t *wq = wqueue + wave * PLO_BIG_QUEUE; uint32_t qn = 0;
            auto drain = [&](uint32_t cntq) {                           // the first min(cntq, 64) queued entries: hashed table (or its fall-back)
                __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier();
                const bool on = lane < cntq;
                const uint32_t raw = on ? __hip_atomic_load(&wq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                if (on) retire_entry((raw & 0x7FFFu) | (((raw >> 15) & 31u) << 16), (raw >> 20) & 31u, (raw >> 25) & 31u, make_uint2(0, 0), make_uint2(0, 0));
                if (cntq > 64u) {
                    const bool mv = lane + 64u < cntq;
                    const uint32_t x_ = mv ? __hip_atomic_load(&wq[lane + 64u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
                    __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier();
                    if (mv) __hip_atomic_store(&wq[lane], x_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier();
            };
#endif
            const uint32_t selsh = l0 == a ? 14u : 20u;                 // the new column's entry carries the +-1 flag and the value index of the l0 entry
            for (uint32_t k0 = 0; k0 < nrw; k0 += 64u) {
                const bool have = k0 + lane < nrw;
                const uint32_t qq = have ? wave + (k0 + lane) * nwaves : wave;
                const uint4 R0 = *(const uint4 *)(aff + 4u * qq);
                const uint32_t Lr = have ? (R0.z & 0x3FFFu) : 0u;
                const uint32_t E = wave_incl_scan(Lr), S = E - Lr, T = RL(E, 63);
                const uint32_t safe = RL(R0.y, 0);
                uint32_t qlo = 0;
                const uint32_t FW = P.fwin;
                for (uint32_t w0 = 0; w0 < T; w0 += FW) {
                    if (w0 && tid == 0) ++sh.nwin;
                    if (lane < 32u) __hip_atomic_store(&fm[lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier();
                    if (have && E >= w0 && E - w0 < FW) wg_or((unsigned long long *)&fm[(E - w0) >> 6], 1ull << (E & 63u));
                    __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier();
                    const uint64_t mreg = __hip_atomic_load(&fm[lane & 31u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t mrl = (uint32_t)mreg, mrh = (uint32_t)(mreg >> 32);
                    const uint32_t ntw = ((T - w0 < FW ? T - w0 : FW) + 63u) >> 6;
                    // Trip t_: the row of each lane (advances qlo), the row's record fetched from the lane that holds it -- four
                    // ds_bpermute in flight together, one wait -- and the request for the entry.  A mark at bit i says "a row ended
                    // before position i": rows ended at or below a lane = bit 0 + v_mbcnt of the mask shifted right by one.
                    auto prep = [&](uint32_t t_, uint32_t &ad_, uint32_t &z_, uint32_t &pp_, uint32_t &rz_, uint32_t &e_) {
                        const bool ok = t_ < ntw;                          // (the trip after the window's last one is requested nowhere and marks nothing)
                        const uint32_t ml = ok ? RL(mrl, t_ & 31u) : 0u, mh = ok ? RL(mrh, t_ & 31u) : 0u;
                        const uint32_t m1l = (ml >> 1) | (mh << 31), m1h = mh >> 1;
                        const int qa = (int)((qlo + (ml & 1u) + __builtin_amdgcn_mbcnt_hi(m1h, __builtin_amdgcn_mbcnt_lo(m1l, 0u))) << 2);
                        qlo += (uint32_t)__builtin_popcount(ml) + (uint32_t)__builtin_popcount(mh);
                        const uint32_t rb_ = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.y), S_ = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)S);
                        pp_ = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.x); rz_ = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.z);
                        const uint32_t f_ = w0 + (t_ << 6) + lane;
                        z_ = f_ - S_; ad_ = rb_ + z_;
                        e_ = ent[ok && f_ < T ? ad_ : safe];
                    };
#if defined(PLO_BIG_PIPETRIP) && !defined(PLO_BIG_PROFILE)
                    // Round 4: the trip as ONE hand-ordered instruction stream.  Measured with a clock behind every LDS round trip (profile
                    // build, profiles/r04_sweep_trip_clocks.txt): per trip the wave waited in turn for the record permutes of the next trip
                    // (~130 cycles), the ratio-identifier lookup (73), the pair read (126) and the compare-and-swap of the few claiming
                    // lanes (240) -- a chain of ~570 cycles of LDS latency in a trip of ~1,700, with 2 to 4 waves per SIMD to hide it.
                    // Here (1) the lookup of THIS trip's identifiers is issued first, (2) the next trip's four permutes behind it, (3) the
                    // pair read as soon as the identifier is back, (4) the next trip's entry request, (5) this trip's stores -- so the
                    // three latencies overlap -- and (6) a claim is OPTIMISTIC: the compare-and-swap is issued, the bitmap bit and the count
                    // are added at once, and its returned word is looked at one trip later; a claim that lost its slot to another key (0.2 %
                    // of the lanes) takes its count back and goes through the probe loop then.  Counts are read by the flush, behind the
                    // barrier that ends the sweep, so a count that sits on a wrong slot for one trip is seen by nobody.
                    uint32_t adc, zc, ppc, rzc, ec; prep(0u, adc, zc, ppc, rzc, ec);
                    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): see PLAINTRIP below
                    uint32_t dk = 0xFFFFFFFFu, dold = 0u, de = 0u, drz = 0u;          // the claim of the trip before: key | second slot << 31, returned word, entry, record bits
                    auto settle = [&]() {
                        const bool lost = dk != 0xFFFFFFFFu && dold != 0xFFFFFFFFu && dold != (dk & 0x7FFFFFFFu);
                        if (__builtin_amdgcn_ballot_w64(lost)) {
                            if (lost) {
                                const uint32_t key = dk & 0x7FFFFFFFu, s = ((uint32_t)__umul24(key, 0x9E3779u) >> (32u - aggbits)) & ~1u;
                                wg_sub(&aggc32[s >> 1], (dk >> 31) ? 0x10000u : 1u);
                                if (!agg_add_rid_bm(aggk, aggc32, aggbits, key, aggbm)) retire_entry(de, (drz >> 15) & 31u, (drz >> 21) & 31u, make_uint2(0, 0), make_uint2(0, 0), true);
                            }
                        }
                        dk = 0xFFFFFFFFu;
                    };
                    for (uint32_t t = 0; t < ntw; ++t) {
                        // (1) this trip's ratio identifiers (idle lanes look up a valid entry too: no branch around the read)
                        const uint32_t cE = PLO_ECOL(ec), viE = PLO_EVI(ec) & 31u, viaE = (rzc >> 15) & 31u;
                        const uint32_t xidE = rtid[cE < a ? (viaE * PLO_RSTRIDE) | viE : (viE * PLO_RSTRIDE) | viaE];
                        PLO_SCHED_FENCE();
                        // (2) the next trip's row records
                        const uint32_t t_ = t + 1u; const bool okn = t_ < ntw;
                        const uint32_t ml = okn ? RL(mrl, t_ & 31u) : 0u, mh = okn ? RL(mrh, t_ & 31u) : 0u;
                        const uint32_t m1l = (ml >> 1) | (mh << 31), m1h = mh >> 1;
                        const int qa = (int)((qlo + (ml & 1u) + __builtin_amdgcn_mbcnt_hi(m1h, __builtin_amdgcn_mbcnt_lo(m1l, 0u))) << 2);
                        qlo += (uint32_t)__builtin_popcount(ml) + (uint32_t)__builtin_popcount(mh);
                        const uint32_t rbn = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.y), Sn = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)S);
                        const uint32_t ppn = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.x), rzn = (uint32_t)__builtin_amdgcn_ds_bpermute(qa, (int)R0.z);
                        PLO_SCHED_FENCE();
                        // (3) this trip's pair of aggregation slots
                        const uint32_t keyE = (cE << PLO_RIDB) | xidE, sE = ((uint32_t)__umul24(keyE, 0x9E3779u) >> (32u - aggbits)) & ~1u;
                        const unsigned long long kk = __hip_atomic_load((unsigned long long *)(aggk + sE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        PLO_SCHED_FENCE();
                        // (4) the next trip's entry
                        const uint32_t fn = w0 + (t_ << 6) + lane, zn = fn - Sn, adn = rbn + zn;
                        const uint32_t en = ent[okn && fn < T ? adn : safe];
                        PLO_SCHED_FENCE();
                        // (5) this trip's row rewrite (:96-110), stores unconditional (idle lanes write their dump word)
                        const uint32_t pa = ppc & 0xFFFFu, pb = ppc >> 16;
                        const bool in = w0 + (t << 6) + lane < T, act = in && zc != pa && zc != pb;
                        ent[act && zc > pa ? adc - 1u - (zc > pb ? 1u : 0u) : dump] = ec;
                        ent[in && zc + 1u == (rzc & 0x3FFFu) ? adc - 1u : dump] = (((rzc >> selsh) & 63u) << 15) | lm;
                        PLO_SCHED_FENCE();
                        // (6) the claims of the trip before, then this trip's keys
                        settle();
                        const uint32_t k0 = (uint32_t)kk, k1 = (uint32_t)(kk >> 32);
                        const bool hit0 = k0 == keyE, hit = hit0 || k1 == keyE, emp0 = k0 == 0xFFFFFFFFu;
                        const bool tryc = act && !hit && (emp0 || k1 == 0xFFFFFFFFu), full = act && !hit && !tryc;
                        const uint32_t sec = hit ? (hit0 ? 0u : 1u) : (emp0 ? 0u : 1u), tE = sE + sec;
                        if (tryc) { dold = wg_cas(&aggk[tE], 0xFFFFFFFFu, keyE); wg_or(&aggbm[tE >> 5], 1u << (tE & 31u)); dk = keyE | (sec << 31); de = ec; drz = rzc; }
                        if (act && !full) wg_add(&aggc32[sE >> 1], sec ? 0x10000u : 1u);
                        if (__builtin_amdgcn_ballot_w64(full)) {              // both slots hold other keys: the probe loop (one trip in ten)
                            if (full) { if (!agg_add_rid_bm(aggk, aggc32, aggbits, keyE, aggbm)) retire_entry(ec, viaE, (rzc >> 21) & 31u, make_uint2(0, 0), make_uint2(0, 0), true); }
                        }
                        adc = adn; zc = zn; ppc = ppn; rzc = rzn; ec = en;
                    }
                    settle();
#elif !defined(PLO_BIG_PREFETCH1) && !defined(PLO_BIG_PROFILE)
                    // Round 4 (default): entries are requested TWO trips ahead and the loop is unrolled three times, so that the three sets
                    // of trip registers (address, position in the row, the row's record words, the entry) rotate by NAME.  The one-ahead loop
                    // below ends in `ec = en`: a copy of the entry requested at the top of the same trip, i.e. a wait for that load at the
                    // bottom of every trip -- a load had one trip body (~1,500 cycles) to come back, and with 512 workgroups in flight it
                    // takes longer (a first two-ahead version that kept the copies gained nothing: the copy of the newest set waited just
                    // the same).  A store of a trip goes at most two positions below the storing entry's own address, which the same or an
                    // earlier trip has loaded: no load of a later trip, however early, sees it.  816 -> 840-846 candidates/s A/B on one box,
                    // a candidate alone 436 -> 421 ms (profiles/r04_ab_config5.txt); three trips ahead (four sets): 832, the registers spill.
                    auto body3 = [&](uint32_t t, uint32_t adc, uint32_t zc, uint32_t ppc, uint32_t rzc, uint32_t ec) {
                        const uint32_t pa = ppc & 0xFFFFu, pb = ppc >> 16;
                        const bool in = w0 + (t << 6) + lane < T, act = in && zc != pa && zc != pb;
                        // (the two UNCONDITIONAL stores: see the one-ahead loop below)
                        ent[act && zc > pa ? adc - 1u - (zc > pb ? 1u : 0u) : dump] = ec;
                        ent[in && zc + 1u == (rzc & 0x3FFFu) ? adc - 1u : dump] = (((rzc >> selsh) & 63u) << 15) | lm;
#ifdef PLO_BIG_DIRECT
                        if (use_direct) {
                            const uint32_t cE = PLO_ECOL(ec), viE = PLO_EVI(ec) & 31u, viaE = (rzc >> 15) & 31u;
                            const bool one = viE == viaE, mone = (rzc >> 31) != 0u && viE == ((rzc >> 26) & 31u);
                            const bool dir = act && cE < P.dcols && (one || mone);
                            if (dir) wg_add(&dcnt[cE], one ? 1u : 0x10000u);
                            const bool qd = act && !dir;
                            const uint64_t qm = __builtin_amdgcn_ballot_w64(qd);
                            if (qm) {
                                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                                if (qd) __hip_atomic_store(&wq[qn + rk], cE | (viE << 15) | (viaE << 20) | (((rzc >> 21) & 31u) << 25), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                qn += (uint32_t)__builtin_popcountll(qm);
                                if (qn >= 64u) { drain(qn); qn -= 64u; }
                            }
                        } else
#endif
                        if (act) retire_entry(ec, (rzc >> 15) & 31u, (rzc >> 21) & 31u, make_uint2(0, 0), make_uint2(0, 0));
                    };
                    uint32_t a0_, z0_, p0_, r0_, e0_, a1_, z1_, p1_, r1_, e1_, a2_, z2_, p2_, r2_, e2_;
                    prep(0u, a0_, z0_, p0_, r0_, e0_); prep(1u, a1_, z1_, p1_, r1_, e1_);
                    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): see the one-ahead loop below
                    for (uint32_t t = 0;;) {
                        prep(t + 2u, a2_, z2_, p2_, r2_, e2_); body3(t, a0_, z0_, p0_, r0_, e0_); if (++t >= ntw) break;
                        prep(t + 2u, a0_, z0_, p0_, r0_, e0_); body3(t, a1_, z1_, p1_, r1_, e1_); if (++t >= ntw) break;
                        prep(t + 2u, a1_, z1_, p1_, r1_, e1_); body3(t, a2_, z2_, p2_, r2_, e2_); if (++t >= ntw) break;
                    }
#else
                    uint32_t adc, zc, ppc, rzc, ec; prep(0u, adc, zc, ppc, rzc, ec);
                    // (the wait counters of the loop header merge both incoming edges: with the first entry still in flight here every
                    // trip would wait for all but one memory operation, i.e. for the stores of the trip before)
                    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
                    for (uint32_t t = 0; t < ntw; ++t) {
                        uint32_t adn, zn, ppn, rzn, en; prep(t + 1u, adn, zn, ppn, rzn, en);
#ifdef PLO_BIG_PROFILE
                        const unsigned long long t0_ = clock64();
#endif
                        const uint32_t pa = ppc & 0xFFFFu, pb = ppc >> 16;
                        const bool in = w0 + (t << 6) + lane < T, act = in && zc != pa && zc != pb;
                        // the row is rewritten in the same pass (:96-110): entries right of the first removed position shift left;
                        // the new column's entry goes last (len and the +-1 counters were updated by the search).  Both stores are
                        // UNCONDITIONAL (idle lanes write to a dump word behind the entries): behind a branch the compiler cannot
                        // count them and waits for every store of the trip (vmcnt(0)) before it uses the entry requested a trip ahead.
                        ent[act && zc > pa ? adc - 1u - (zc > pb ? 1u : 0u) : dump] = ec;
                        ent[in && zc + 1u == (rzc & 0x3FFFu) ? adc - 1u : dump] = (((rzc >> selsh) & 63u) << 15) | lm;
#ifdef PLO_BIG_PROFILE
                        __builtin_amdgcn_wave_barrier(); const unsigned long long t1_ = clock64(); probe_iters = 0;
#endif
#if defined(PLO_BIG_DIRECT) && !defined(PLO_BIG_PROFILE)
                        if (use_direct) {
                            const uint32_t cE = PLO_ECOL(ec), viE = PLO_EVI(ec) & 31u, viaE = (rzc >> 15) & 31u;
                            const bool one = viE == viaE, mone = (rzc >> 31) != 0u && viE == ((rzc >> 26) & 31u);
                            const bool dir = act && cE < P.dcols && (one || mone);
                            if (dir) wg_add(&dcnt[cE], one ? 1u : 0x10000u);
                            const bool qd = act && !dir;
                            const uint64_t qm = __builtin_amdgcn_ballot_w64(qd);
                            if (qm) {
                                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                                if (qd) __hip_atomic_store(&wq[qn + rk], cE | (viE << 15) | (viaE << 20) | (((rzc >> 21) & 31u) << 25), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                qn += (uint32_t)__builtin_popcountll(qm);          // (below 64 before: never beyond the queue's 128 words)
                                if (qn >= 64u) { drain(qn); qn -= 64u; }
                            }
                        } else
#endif
                        if (act) retire_entry(ec, (rzc >> 15) & 31u, (rzc >> 21) & 31u, make_uint2(0, 0), make_uint2(0, 0));
#ifdef PLO_BIG_PROFILE
                        {   __builtin_amdgcn_wave_barrier(); const unsigned long long t2_ = clock64(); pw0 += t1_ - t0_; pw1 += t2_ - t1_; pw2 += t0_ - tl_; tl_ = t2_; ++ptr;
                            uint32_t mx = probe_iters; for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)mx, o); mx = u > mx ? u : mx; }
                            pit += mx; pact += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act)); tl_ = clock64(); }
#endif
                        adc = adn; zc = zn; ppc = ppn; rzc = rzn; ec = en;
                    }
#endif
                }
            }
#ifdef PLO_BIG_DIRECT
            if (qn) drain(qn);                                          // what the wave's queue still holds
#endif
            } else
#endif
            for (uint32_t k0 = 0; k0 < nrw; k0 += 64u) {
                // two rows per trip; the first chunks of the next pair are requested before the current pair is worked on
                const bool have = k0 + lane < nrw;
                const uint32_t qq = have ? wave + (k0 + lane) * nwaves : wave;
                uint32_t Rpp, Rbase, RL_, Rea, Reb;
                if constexpr (MODE == 2) {
                    const uint4 R0 = *(const uint4 *)(aff + 4u * qq);
                    Rpp = R0.x; Rbase = R0.y; RL_ = have ? (R0.z & 0x3FFFu) : 0u;
                    Rea = ((R0.z >> 14) & 63u) << 15; Reb = ((R0.z >> 20) & 63u) << 15;   // value index and +-1 flag: all the sweep needs of the two entries
                } else {
                    const uint4 R0 = *(const uint4 *)(aff + 8u * qq); const uint2 R1 = *(const uint2 *)(aff + 8u * qq + 4u);
                    Rpp = R0.y; Rbase = R0.z; RL_ = have ? R0.w : 0u; Rea = R1.x; Reb = R1.y;
                }
                const uint32_t cnt = nrw - k0 < 64u ? nrw - k0 : 64u;
                uint32_t baseA = RL(Rbase, 0), LA = RL(RL_, 0), baseB = RL(Rbase, 1), LB = RL(RL_, 1);
                uint32_t eA = ent[baseA + (lane < LA ? lane : 0u)], eB = ent[baseB + (lane < LB ? lane : 0u)];
                for (uint32_t kk = 0; kk < cnt; kk += 2u) {
#ifdef PLO_BIG_PROFILE
                    unsigned long long tp = clock64();
#endif
                    const uint32_t k2 = (kk + 2u) & 63u, k3 = (kk + 3u) & 63u; const bool morep = kk + 2u < 64u;
                    const uint32_t nbaseA = RL(Rbase, k2), nLA = morep ? RL(RL_, k2) : 0u, nbaseB = RL(Rbase, k3), nLB = morep ? RL(RL_, k3) : 0u;
                    const uint32_t nfA = ent[nbaseA + (lane < nLA ? lane : 0u)], nfB = ent[nbaseB + (lane < nLB ? lane : 0u)];
                    const uint32_t ppA = RL(Rpp, kk), ppB = RL(Rpp, kk + 1u);
                    const uint32_t paA = ppA & 0xFFFFu, pbA = ppA >> 16, paB = ppB & 0xFFFFu, pbB = ppB >> 16;
                    const uint32_t eaA = RL(Rea, kk), ebA = RL(Reb, kk), eaB = RL(Rea, kk + 1u), ebB = RL(Reb, kk + 1u);
                    uint2 VaA = make_uint2(0, 0), VbA = VaA, VaB = VaA, VbB = VaA;
                    if constexpr (MODE != 2) { VaA = VT(PLO_EVI(eaA)); VbA = VT(PLO_EVI(ebA)); VaB = VT(PLO_EVI(eaB)); VbB = VT(PLO_EVI(ebB)); }
                    const uint32_t Lmax = LA > LB ? LA : LB;
                    for (uint32_t z0 = 0; z0 < Lmax; z0 += 64u) {
                        const uint32_t z = z0 + lane, zn = z + 64u;
                        const bool morez = z0 + 64u < Lmax;                       // wave-uniform: rows of one chunk issue no further load (and wait for none)
                        uint32_t neA = 0, neB = 0;
                        if (morez) { neA = ent[baseA + (zn < LA ? zn : 0u)]; neB = ent[baseB + (zn < LB ? zn : 0u)]; }
                        __builtin_amdgcn_wave_barrier();
#ifdef PLO_BIG_PROFILE
                        const unsigned long long t0_ = clock64();
#endif
                        const bool actA = z < LA && z != paA && z != pbA, actB = z < LB && z != paB && z != pbB;
                        // the row is rewritten in the same pass (:96-110): entries right of the first removed position shift left
                        if (actA && z > paA) ent[baseA + z - 1u - (z > pbA ? 1u : 0u)] = eA;
                        if (actB && z > paB) ent[baseB + z - 1u - (z > pbB ? 1u : 0u)] = eB;
#ifdef PLO_BIG_PROFILE
                        const unsigned long long t1_ = clock64();
#endif
                        if (actA) retire_entry(eA, PLO_EVI(eaA), PLO_EVI(ebA), VaA, VbA);
                        if (actB) retire_entry(eB, PLO_EVI(eaB), PLO_EVI(ebB), VaB, VbB);
                        __builtin_amdgcn_wave_barrier();
#ifdef PLO_BIG_PROFILE
                        { const unsigned long long t2_ = clock64(); pw0 += t1_ - t0_; pw1 += t2_ - t1_; pw2 += t0_ - tl_; tl_ = t2_; ++ptr; }
#endif
                        if (morez) { eA = neA; eB = neB; }
                    }
                    if (lane == 0) {                       // the new column's entry goes last (len and the +-1 counters were updated by the search)
                        if (LA) ent[baseA + LA - 2u] = (((l0 == a) ? eaA : ebA) & 0xFFFF8000u) | lm;
                        if (LB) ent[baseB + LB - 2u] = (((l0 == a) ? eaB : ebB) & 0xFFFF8000u) | lm;
                    }
                    baseA = nbaseA; LA = nLA; baseB = nbaseB; LB = nLB; eA = nfA; eB = nfB;
                }
            }
#ifdef PLO_BIG_PROFILE
            if (lane == 0 && M >= 256u) { atomicAdd(&sh.pw[0], pw0); atomicAdd(&sh.pw[1], pw1); atomicAdd(&sh.pw[2], pw2); atomicAdd(&sh.pw[3], ptr); atomicAdd(&sh.pw[4], clock64() - ts_); atomicAdd(&sh.pw[5], 1ull); atomicAdd(&sh.pw[6], pit); atomicAdd(&sh.pw[7], pact); atomicAdd(&sh.pw[8], pq[0]); atomicAdd(&sh.pw[9], pq[1]); atomicAdd(&sh.pw[10], pq[2]); atomicAdd(&sh.pw[14], prt); }
#endif
        }
#undef RL
        BSYNC();
#ifdef PLO_BIG_PROFILE
        if (tid == 0) { const int cl = M >= 256u ? 0 : M >= 64u ? 1 : M >= 16u ? 2 : 3; sh.tb1[cl] += wall_clock64() - tstamp; ++sh.nb[cl]; }
#endif
        PLO_STAMP(3);
        if constexpr (DEFER) {
            // Flush (DEFER), ONE pass over the aggregation table: an entry (c, x) x d retires two triples and creates one.  A triple
            // that may be hot (Bloom filter, then the hot table) is updated there, exactly; everything else is a log record.
            const uint32_t invr = sh.invr, theta = sh.theta;
            auto retired = [&](uint64_t k, uint32_t d, uint32_t o) {
                if (o < d) { { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 201u); } return; }
                wg_sub(&hist[o], 1u); if (o > d) wg_add(&hist[o - d], 1u);
                if (o == M) { wg_sub(&cntM[(uint32_t)(k >> abits)], 1u); wg_sub(&sh.cblk[(uint32_t)(k >> abits) >> 6], 1u); }
            };
            auto inserted = [&](uint64_t k, uint32_t o, uint32_t d) {
                const uint32_t nc = o + d;
                if (nc > P.maxf0 || nc > M) { wg_max(&sh.errflag, (uint32_t)BERR_FREQ); return; }
                if (o > 0u) wg_sub(&hist[o], 1u);
                wg_add(&hist[nc], 1u);
                if (o < theta && nc >= theta) { uint32_t idx = wg_add(&sh.hlcount, 1u); if (idx < P.hlcap) HL[idx] = k; else sh.hlbad = 1u; }
                if (nc == M) {
                    wg_add(&cntM[(uint32_t)(k >> abits)], 1u); wg_add(&sh.cblk[(uint32_t)(k >> abits) >> 6], 1u);
                    uint32_t idx = wg_add(&sh.dmcount, 1u);
                    if (idx < P.dmcap) DM[idx] = k; else wg_max(&sh.errflag, (uint32_t)BERR_DM);
                }
            };
            if (tid == 0) {                                                   // the chosen triple loses its M instances (it is of level M >= theta: hot)
                uint64_t v; const uint32_t sl = hot_slot(tab, key, hbits, v);
                if (sl == 0xFFFFFFFFu || (uint32_t)(v & PLO_GVMASK) != M) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 202u); }
                else { gstore64(&tab[sl], v - (uint64_t)M); retired(key, M, M); }
            }
            const uint32_t nent = sh.aggn, nslot = FAST ? (1u << aggbits) / 16u : nent <= listcap ? nent : (1u << aggbits);     // few entries: walk the slot list, not the table (FAST: the bitmap, 16 slots per thread and round)
            // FAST: behind the bitmap words of the hashed table, the words of the direct table (columns below the new one) when the step's sweep used it
            // (its non-zero words are first listed -- a light pass, 16-bit column numbers in the waves' queue space, idle by now -- so that the
            // rounds below are as many as the touched columns need, not as many as the table has; a list that overflows: the whole table)
#ifdef PLO_BIG_DIRECT
            uint32_t ndir = 0; bool dlisted = false;
            uint16_t *dlist = (uint16_t *)wqueue;
            if constexpr (FAST) {
                if (P.dcols != 0u && naff >= PLO_BIG_DIRECT_MIN) {
                    const uint32_t nd0 = lm < P.dcols ? lm : P.dcols;
                    for (uint32_t w0 = 0; w0 < nd0; w0 += nth) {
                        const uint32_t w = w0 + tid;
                        const bool nz = w < nd0 && dcnt[w] != 0u;
                        const uint64_t bm_ = __builtin_amdgcn_ballot_w64(nz);
                        if (bm_) {
                            uint32_t base_ = 0;
                            if (lane == 0) base_ = wg_add(&sh.dn, (uint32_t)__builtin_popcountll(bm_));
                            base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_);
                            const uint32_t ix = base_ + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm_, 0u));
                            if (nz && ix < 2u * PLO_BIG_QUEUE_WORDS) dlist[ix] = (uint16_t)w;
                        }
                    }
                    BSYNC();
                    const uint32_t dn = sh.dn;
                    dlisted = dn <= 2u * PLO_BIG_QUEUE_WORDS;
                    ndir = dlisted ? dn : nd0;
                }
            }
#endif
            uint32_t nhot = 0;
#ifndef PLO_BIG_DIRECT
            const uint32_t ndir = 0;
#endif
            for (uint32_t s0 = 0; s0 < nslot + ndir; s0 += nth) {
              const uint32_t e_ = s0 + tid;
              uint32_t bits_ = 0; PLO_DIRECT_ONLY(uint32_t dw_ = 0; uint32_t dcol_ = 0; const bool isdir = e_ >= nslot;)
              if constexpr (FAST) {
                  if (e_ < nslot) bits_ = (aggbm[e_ >> 1] >> ((e_ & 1u) << 4)) & 0xFFFFu;
#ifdef PLO_BIG_DIRECT
                  else if (e_ - nslot < ndir) { dcol_ = dlisted ? (uint32_t)dlist[e_ - nslot] : e_ - nslot; dw_ = dcnt[dcol_]; if (dw_) dcnt[dcol_] = 0u; bits_ = ((dw_ & 0xFFFFu) ? 1u : 0u) | ((dw_ >> 16) ? 2u : 0u); }
#endif
              }
              for (bool more_ = true; more_;) {
                bool valid; uint32_t s;
#ifdef PLO_BIG_DIRECT
                if constexpr (FAST) { valid = bits_ != 0u; s = valid ? (isdir ? (uint32_t)__builtin_ctz(bits_) : e_ * 16u + (uint32_t)__builtin_ctz(bits_)) : 0u; bits_ &= bits_ - 1u; }
#else
                if constexpr (FAST) { valid = bits_ != 0u; s = valid ? e_ * 16u + (uint32_t)__builtin_ctz(bits_) : 0u; bits_ &= bits_ - 1u; }
#endif
                else { valid = e_ < nslot; s = valid ? (nent <= listcap ? (uint32_t)agglist[e_] : e_) : 0u; }
                uint32_t c = 0, x = 0, y = 0, d = 0;
                if constexpr (MODE == 2) {
#ifdef PLO_BIG_DIRECT
                    if (FAST && isdir) {
                        if (valid) {
                            const uint32_t xid = s ? P.id_mone : P.id_one;
                            d = s ? dw_ >> 16 : dw_ & 0xFFFFu; c = dcol_;
                            x = rval[xid]; y = rval[c > a ? (uint32_t)invid[xid] : xid];
                        }
                    } else
#endif
                    {
                    const uint32_t kq = valid ? aggk[s] : 0xFFFFFFFFu;
                    valid = kq != 0xFFFFFFFFu;
                    if (valid) {
                        d = aggc16[s]; aggk[s] = 0xFFFFFFFFu; aggc16[s] = 0; c = kq >> PLO_RIDB;
                        const uint32_t xid = kq & ((1u << PLO_RIDB) - 1u);
                        x = rval[xid]; y = rval[c > a ? (uint32_t)invid[xid] : xid];
                    }
                    }
                } else {
                    const uint64_t v = valid ? agg[s] : AEMPTY;
                    valid = v != AEMPTY;
                    if (valid) {
                        agg[s] = AEMPTY;
                        uint64_t k = v >> acb; d = (uint32_t)(v & ((1ull << acb) - 1ull));
                        if (P.agg_dual) { y = (uint32_t)(k & ((1ull << rb) - 1ull)); k >>= rb; }
                        c = (uint32_t)(k >> rb); x = (uint32_t)(k & ((1ull << rb) - 1ull));
                        if (!P.agg_dual) y = c > a ? (P.invtab ? P.invtab[x] : binv(x, p, mu, mers)) : x;   // v_a / v_c
                    }
                }
                uint64_t k1 = 0, k2 = 0, k3 = 0;
                if (valid) {
                    const uint32_t ry = bmul(r, y, p, mu, mers);                        // v_b / v_c
                    const uint32_t x2 = c < b ? ry : bmul(x, invr, p, mu, mers);
                    k1 = c < a ? BKEY(c, a, x) : BKEY(a, c, x); k2 = c < b ? BKEY(c, b, x2) : BKEY(b, c, x2);
                    k3 = BKEY(c, lm, l0 == a ? y : ry);
                }
                bool c1 = valid, c2 = valid;                                            // still to be logged
                if (valid && dbloom_test(bloom, k1)) {
                    uint64_t v; const uint32_t sl = hot_slot(tab, k1, hbits, v);
                    if (sl != 0xFFFFFFFFu) { const uint32_t o = (uint32_t)(v & PLO_GVMASK); if (o >= d) gstore64(&tab[sl], v - (uint64_t)d); retired(k1, d, o); c1 = false; ++nhot; }   // ONE writer per slot in this pass: a plain store
                }
                if (valid && dbloom_test(bloom, k2)) {
                    uint64_t v; const uint32_t sl = hot_slot(tab, k2, hbits, v);
                    if (sl != 0xFFFFFFFFu) { const uint32_t o = (uint32_t)(v & PLO_GVMASK); if (o >= d) gstore64(&tab[sl], v - (uint64_t)d); retired(k2, d, o); c2 = false; ++nhot; }
                }
                const bool h3 = valid && d >= theta, c3 = valid && !h3 && d >= 2u;       // (seen once in its only step: frequency 1 for ever, not kept)
                if (h3) {
                    const uint32_t o = hot_addn(tab, k3, d, hbits, &sh.hotn);
                    if (o != 0u) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 203u); } else { inserted(k3, 0u, d); dbloom_set(bloom, k3); ++nhot; }
                }
                dlog_append3(dlog, &sh.logn, P.logcap, c1, PLO_DREC(k1, d, false), c2, PLO_DREC(k2, d, false), c3, PLO_DREC(k3, d, true), &sh.errflag);
                if constexpr (FAST) more_ = __builtin_amdgcn_ballot_w64(bits_ != 0u) != 0ull; else more_ = false;
              }
            }
            BSYNC();
            // entries that found no room in the LDS table: their pairs with the new column were not summed, so the hot table sums
            // them (exact whatever the total turns out to be; the next merge sends totals below the window back to the store)
            const uint32_t nsp = sh.nspill < spillcap ? sh.nspill : spillcap;
            for (uint32_t e = tid; e < nsp; e += nth) {
                const uint64_t k = spill[e];
                const uint32_t o = hot_addn(tab, k, 1u, hbits, &sh.hotn);
                if (o == 0xFFFFFFFFu) { { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); wg_max(&sh.derr, 204u); } continue; }
                inserted(k, o, 1u); dbloom_set(bloom, k); ++nhot;
            }
            if (nhot) wg_add(&sh.hotops, nhot);
            PLO_STAMP(4);
        } else {
        // Flush, first pass: the summed retirements, one table atomic per distinct triple; the triple itself loses all its M
        // instances.  The pair an entry forms with the NEW column has the same multiplicity as its retirements and its
        // ratio coeff/v_c is a function of (c, x) as well (x or 1/x, times r when coeff = v_b): the entry is rewritten in
        // place to (c, that ratio) and the second pass inserts it -- the rows are not hashed a second time.
        {
            const uint32_t invr = sh.invr;
            auto retired = [&](uint64_t k, uint32_t d, uint32_t o) {       // bookkeeping of a retirement that found frequency o
                if (o == 0u) return;                                       // not in the table: a triple of frequency 1 (pruned)
                if (o < d) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); return; }
                wg_sub(&hist[o], 1u); if (o > d) wg_add(&hist[o - d], 1u);
                if (o == M) { wg_sub(&cntM[(uint32_t)(k >> abits)], 1u); wg_sub(&sh.cblk[(uint32_t)(k >> abits) >> 6], 1u); }
            };
            if (tid == 0) retired(key, M, gtab_subn(tab, key, M, hbits));
            const uint32_t nent = sh.aggn, nslot = nent <= listcap ? nent : (1u << aggbits);
            // Every lane runs its own entries: it takes the next one as soon as both keys of the current one are settled, so
            // a probe round costs the wave one memory round trip whatever the other lanes' probe lengths are.  The keys of
            // the entries are all different (and different from the chosen triple), so every slot has ONE writer in this
            // pass: the new frequency is a plain store of the word the probe has just read -- no read-modify-write atomic
            // (tests/micro/random_access.hip: the chip sustains 27 G table atomics/s however local they are, against
            // 50-900 G loads/s; 1.2e7 of them per candidate were a third of the step).
            {
                const uint32_t tmask = (1u << hbits) - 1u; const bool bylist = nent <= listcap;
                uint32_t e = tid; bool busy = false, p1 = false, p2 = false;
                uint64_t k1 = 0, k2 = 0; uint32_t d = 0, s1 = 0, s2 = 0;
#ifdef PLO_BIG_PROFILE
                unsigned long long fq0 = 0, fq1 = 0, fq2 = 0, fqn = 0; const unsigned long long fts = clock64(); uint32_t nfl1 = 0;
#endif
                for (;;) {
#ifdef PLO_BIG_PROFILE
                    const unsigned long long ft0 = clock64();
#endif
                    if (!busy) {
                        while (e < nslot) {
                            const uint32_t s = bylist ? (uint32_t)agglist[e] : e; e += nth;
                            uint32_t c, x, y = 0;
                            if constexpr (MODE == 2) {
                                const uint32_t kq = aggk[s];
                                if (kq == 0xFFFFFFFFu) continue;
                                d = aggc16[s]; c = kq >> PLO_RIDB;
                                const uint32_t xid = kq & ((1u << PLO_RIDB) - 1u);
                                x = rval[xid]; y = rval[c > a ? (uint32_t)invid[xid] : xid];
                            } else {
                                const uint64_t v = agg[s];
                                if (v == AEMPTY) continue;
                                uint64_t k = v >> acb; d = (uint32_t)(v & ((1ull << acb) - 1ull));
                                if (P.agg_dual) { y = (uint32_t)(k & ((1ull << rb) - 1ull)); k >>= rb; }
                                c = (uint32_t)(k >> rb); x = (uint32_t)(k & ((1ull << rb) - 1ull));
                                if (!P.agg_dual) y = c > a ? (P.invtab ? P.invtab[x] : binv(x, p, mu, mers)) : x;   // v_a / v_c
                            }
                            const uint32_t ry = bmul(r, y, p, mu, mers);                        // v_b / v_c
                            const uint32_t x2 = c < b ? ry : bmul(x, invr, p, mu, mers);
                            k1 = c < a ? BKEY(c, a, x) : BKEY(a, c, x); k2 = c < b ? BKEY(c, b, x2) : BKEY(b, c, x2);
                            s1 = ghash(k1, hbits); s2 = ghash(k2, hbits); p1 = p2 = true; busy = true;
                            if constexpr (MODE != 2) agg[s] = (((((uint64_t)c) << rb) | (l0 == a ? y : ry)) << PLO_GVB) | d;    // (c, coeff / v_c), same count (mode 2: the second pass derives it again)
#ifdef PLO_BIG_PROFILE
                            ++nfl1;
#endif
                            break;
                        }
                    }
                    if (!__builtin_amdgcn_ballot_w64(busy)) break;
#ifdef PLO_BIG_PROFILE
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long ft1 = clock64();
#endif
                    uint64_t v1[PLO_GWIN], v2[PLO_GWIN];
#pragma unroll
                    for (uint32_t j = 0; j < PLO_GWIN; ++j) {
                        v1[j] = (busy && p1) ? gload64(&tab[(s1 + j) & tmask]) : 0ull;
                        v2[j] = (busy && p2) ? gload64(&tab[(s2 + j) & tmask]) : 0ull;
                    }
#ifdef PLO_BIG_PROFILE
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long ft2 = clock64();
#endif
                    if (busy) {
                        if (p1) {
                            uint32_t adv = PLO_GWIN;
#pragma unroll
                            for (uint32_t j = 0; j < PLO_GWIN; ++j) if (p1) {
                                if ((v1[j] >> PLO_GVB) == k1) {
                                    p1 = false; adv = j; const uint32_t o = (uint32_t)(v1[j] & PLO_GVMASK);
                                    if (o >= d) gstore64(&tab[(s1 + j) & tmask], v1[j] - (uint64_t)d);
                                    retired(k1, d, o);
                                } else if (v1[j] == PLO_GEMPTY) { p1 = false; adv = j; }                  // not in the table: a triple of frequency 1 (pruned)
                            }
                            s1 = (s1 + adv) & tmask;
                        }
                        if (p2) {
                            uint32_t adv = PLO_GWIN;
#pragma unroll
                            for (uint32_t j = 0; j < PLO_GWIN; ++j) if (p2) {
                                if ((v2[j] >> PLO_GVB) == k2) {
                                    p2 = false; adv = j; const uint32_t o = (uint32_t)(v2[j] & PLO_GVMASK);
                                    if (o >= d) gstore64(&tab[(s2 + j) & tmask], v2[j] - (uint64_t)d);
                                    retired(k2, d, o);
                                } else if (v2[j] == PLO_GEMPTY) { p2 = false; adv = j; }
                            }
                            s2 = (s2 + adv) & tmask;
                        }
                        busy = p1 || p2;
                    }
#ifdef PLO_BIG_PROFILE
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); { const unsigned long long ft3 = clock64(); fq0 += ft1 - ft0; fq1 += ft2 - ft1; fq2 += ft3 - ft2; ++fqn; }
#endif
                }
#ifdef PLO_BIG_PROFILE
                wg_add(&sh.fl1, nfl1);
                if (lane == 0 && M >= 256u) { atomicAdd(&sh.pw[8], fq0); atomicAdd(&sh.pw[9], fq1); atomicAdd(&sh.pw[10], fq2); atomicAdd(&sh.pw[11], fqn); atomicAdd(&sh.pw[12], clock64() - fts); atomicAdd(&sh.pw[13], 1ull); }
#endif
            }
        }
        PLO_BIG_FENCE(); BSYNC();
        PLO_STAMP(4);
        if (sh.errflag) break;
        // Flush, second pass: the insertions (:132-142), after every retirement (dead-slot reuse needs that order)
        {
            auto inserted = [&](uint64_t k, uint32_t o, uint32_t d) {
                const uint32_t nc = o + d;
                if (nc > P.maxf0 || nc > M) { wg_max(&sh.errflag, (uint32_t)BERR_FREQ); return; }
                if (o > 0u) wg_sub(&hist[o], 1u);
                wg_add(&hist[nc], 1u);
                if (o < sh.theta && nc >= sh.theta) { uint32_t idx = wg_add(&sh.hlcount, 1u); if (idx < P.hlcap) HL[idx] = k; else sh.hlbad = 1u; }
                if (nc == M) {
                    wg_add(&cntM[(uint32_t)(k >> abits)], 1u); wg_add(&sh.cblk[(uint32_t)(k >> abits) >> 6], 1u);
                    uint32_t idx = wg_add(&sh.dmcount, 1u);
                    if (idx < P.dmcap) DM[idx] = k; else wg_max(&sh.errflag, (uint32_t)BERR_DM);
                }
            };
            const uint32_t nent = sh.aggn, nslot = nent <= listcap ? nent : (1u << aggbits);
            for (uint32_t e0 = tid; e0 < nslot; e0 += PLO_FLU * nth) {
                uint64_t kk[PLO_FLU]; uint32_t dd[PLO_FLU], oo[PLO_FLU]; bool lv[PLO_FLU];
#pragma unroll
                for (int u = 0; u < (int)PLO_FLU; ++u) {
                    const uint32_t e = e0 + (uint32_t)u * nth; bool ok = e < nslot;
                    const uint32_t s = ok ? (nent <= listcap ? (uint32_t)agglist[e] : e) : 0u;
                    uint64_t k = 0; uint32_t d = 0;
                    if constexpr (MODE == 2) {
                        const uint32_t kq = ok ? aggk[s] : 0xFFFFFFFFu;
                        ok = kq != 0xFFFFFFFFu;
                        if (ok) {
                            d = aggc16[s]; aggk[s] = 0xFFFFFFFFu; aggc16[s] = 0;
                            const uint32_t c = kq >> PLO_RIDB, xid = kq & ((1u << PLO_RIDB) - 1u), y = rval[c > a ? (uint32_t)invid[xid] : xid];
                            k = BKEY(c, lm, l0 == a ? y : bmul(r, y, p, mu, mers));
                        }
                    } else {
                        const uint64_t v = ok ? agg[s] : AEMPTY;
                        ok = v != AEMPTY;                                    // (an entry rewritten by the first pass has a count below 2^acb in its low 16 bits: never this pattern)
                        if (ok) {
                            agg[s] = AEMPTY;
                            const uint64_t kc = v >> PLO_GVB; d = (uint32_t)(v & PLO_GVMASK);
                            k = BKEY((uint32_t)(kc >> rb), lm, (uint32_t)(kc & ((1ull << rb) - 1ull)));
                        }
                    }
                    if (d < 2u && P.prune) ok = false;                       // seen once in its only step: frequency 1 for ever, never chosen, not kept
                    kk[u] = k; dd[u] = d; lv[u] = ok;
#ifdef PLO_BIG_PROFILE
                    if (ok) wg_add(&sh.fl2, 1u);
#endif
                }
                gtab_addnN<PLO_FLU>(tab, kk, dd, lv, hbits, oo);
#pragma unroll
                for (int u = 0; u < (int)PLO_FLU; ++u) if (lv[u]) {
                    if (oo[u] == 0xFFFFFFFFu) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); continue; }
                    inserted(kk[u], oo[u], dd[u]);
                }
            }
            const uint32_t nsp = sh.nspill < spillcap ? sh.nspill : spillcap;      // entries that found no room in the LDS table
            for (uint32_t e = tid; e < nsp; e += nth) {
                const uint64_t k = spill[e];
                const uint32_t o = gtab_addn(tab, k, 1u, hbits);
                if (o == 0xFFFFFFFFu) { wg_max(&sh.errflag, (uint32_t)BERR_TABLE); continue; }
                inserted(k, o, 1u);
            }
        }
        }
        PLO_STAMP(6);
        BSYNC();
        PLO_STAMP(5);
        // row list of the new column, multiplier reuse (:153-169), counters
        {
            if (tid == 0) { ncrptr[lm - n + 1u] = ncrptr[lm - n] + naff; clen[lm] = naff; sh.part[0] = 0; }      // (the row search wrote the rows)
        }
        BSYNC();
        if (!P.unit) {
            const uint32_t rho = swap ? sh.invr : r, asgs = babs(rho, p);
            if (!babsone(asgs, p)) {
                const uint32_t nm = sh.nmult; bool hit = false;
                for (uint32_t k = tid; k < nm; k += nth) hit |= (multc[k] == l1 && multv[k] == asgs);
                if (hit) sh.part[0] = 1u;
                BSYNC();
                if (tid == 0 && !sh.part[0]) {
                    if (nm >= P.multcap) wg_max(&sh.errflag, (uint32_t)BERR_MULT);
                    else { multc[nm] = l1; multv[nm] = asgs; sh.nmult = nm + 1u; ++sh.nbmul; }
                }
            }
        }
        if (tid == 0) { ++sh.nbadd; ++sh.steps; sh.ncols = lm + 1u; sh.spilltot += sh.nspill; if (sh.aggn > listcap) ++sh.listover; }      // :292, :190-191
        PLO_BIG_FENCE(); BSYNC();
        PLO_STAMP(7);
    }
    BSYNC();
    if (sh.errflag) { if (tid == 0) atomicMax(errw, sh.errflag); return 0; }
    return 1;   // CSE phase done; ProgramGen follows in the caller
#undef BKEY
}


#ifndef PLO_TRI_R
#define PLO_TRI_R 4u               /* Triangle: registers per lane holding the rows of a column (64 rows each) */
#endif
#define PLO_PGFLAG 0x8000u
#define PLO_PGCNT 0x7FFFu
#define PLO_BFRESH 0xFFFFFFFFu

// ProgramGen at scale, counts only (reference :513-611; counting rules derived in
// plo_cse_wave.hip).  Workgroup-parallel for the factoring passes and the output
// count; the Triangle pass walks, with one wave, only the columns that still hold
// two or more non +-1 entries (after FactorOutColumns those have pairwise distinct
// |values|, so they are few and short).
__device__ __forceinline__ uint64_t big_program_gen(const BigPlan &P, uint8_t *ws, BigShared &sh, uint32_t *scratch, uint32_t *errw)
{
    // (the {value, inverse} table is read from global memory here: `scratch` may overlap its LDS copy)
    const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63u, wave = tid >> 6, nwaves = nth >> 6;
    uint64_t *tab   = (uint64_t *)(ws + P.o_tab);
    uint32_t *col   = (uint32_t *)(ws + P.o_col), *val = (uint32_t *)(ws + P.o_val), *inv = (uint32_t *)(ws + P.o_inv);
    uint32_t *len   = (uint32_t *)(ws + P.o_len);
    uint32_t *multc = (uint32_t *)(ws + P.o_multc), *multv = (uint32_t *)(ws + P.o_multv);
    uint32_t *tcnt  = (uint32_t *)(ws + P.o_tcnt), *tptr2 = (uint32_t *)(ws + P.o_tptr2), *tlist = (uint32_t *)(ws + P.o_tlist), *cols2 = (uint32_t *)(ws + P.o_cols2);
    const uint32_t p = P.p, rb = P.rb, m = P.m, ncols0 = sh.ncols, mers = P.mers;
    const uint64_t mu = P.mu;

    if (tid == 0) { sh.acc0 = 0; sh.acc1 = 0; sh.errflag = 0; }
    BSYNC();
    if (P.unit) {                                                      // all +-1: len-1 additions per row (:576)
        uint32_t acc = 0;
        for (uint32_t i = tid; i < m; i += nth) { uint32_t L = len[i]; acc += L > 1u ? L - 1u : 0u; }
        if (acc) wg_add(&sh.acc0, acc);
        BSYNC();
        return ((uint64_t)(sh.nbadd + sh.acc0) << 32) | sh.nbmul;
    }
    // ProgramGen creates values (FactorOutRows sums, Triangle quotients): expand the packed rows to col / val / inv
    {
        const uint32_t *ent = (const uint32_t *)(ws + P.o_ent);
        for (uint32_t i = wave; i < m; i += nwaves) {
            const uint32_t base = P.rs[i], L = len[i];
            for (uint32_t z = lane; z < L; z += 64u) { const uint32_t e = ent[base + z]; const uint2 V = P.vt[PLO_EVI(e)]; col[base + z] = PLO_ECOL(e); val[base + z] = V.x; inv[base + z] = V.y; }
        }
    }
    PLO_BIG_FENCE(); BSYNC();
    // table region for the (column,|v|) multiset, sized by the live entries
    {
        uint32_t acc = 0;
        for (uint32_t i = tid; i < m; i += nth) acc += len[i];
        if (acc) wg_add(&sh.acc0, acc);
        BSYNC();
    }
    const uint32_t live = sh.acc0 + sh.nmult;
    uint32_t hb = 6; while ((1ull << hb) < 4ull * live + 64ull && hb < P.hbits) ++hb;
    if ((1ull << hb) < 2ull * live + 16ull) { if (tid == 0) { atomicMax(errw, (uint32_t)BERR_PGEN); sh.errflag = BERR_PGEN; } BSYNC(); return 0; }
    for (uint64_t s = tid; s < (1ull << hb); s += nth) tab[s] = PLO_GEMPTY;
    BSYNC();
    if (tid == 0) sh.acc0 = 0;
    PLO_BIG_FENCE(); BSYNC();
    for (uint32_t k = tid; k < sh.nmult; k += nth)
        if (!gtab_flag(tab, ((uint64_t)multc[k] << rb) | multv[k], PLO_PGFLAG, hb)) wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
    PLO_BIG_FENCE(); BSYNC();
    // A1 occurrences of (j,e)
    for (uint32_t i = wave; i < m; i += nwaves) {
        const uint32_t base = P.rs[i], L = len[i];
        for (uint32_t z = lane; z < L; z += 64u) {
            const uint32_t e = babs(val[base + z], p);
            if (!babsone(e, p)) if (!gtab_add(tab, ((uint64_t)col[base + z] << rb) | e, 1u, hb)) wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
        }
    }
    PLO_BIG_FENCE(); BSYNC();
    // A2 one multiplication per repeated (j,e) not yet in multiples (:335-352)
    {
        uint32_t cnt = 0;
        for (uint64_t s = tid; s < (1ull << hb); s += nth) {
            uint64_t v = gload64(&tab[s]);
            if (v != PLO_GEMPTY && ((uint32_t)v & PLO_PGCNT) >= 2u && !((uint32_t)v & PLO_PGFLAG)) { ++cnt; wg_or((unsigned long long *)&tab[s], (unsigned long long)PLO_PGFLAG); }
        }
        if (cnt) wg_add(&sh.acc1, cnt);
    }
    PLO_BIG_FENCE(); BSYNC();
    if (tid == 0) { sh.nbmul += sh.acc1; sh.acc1 = 0; }
    // A3 repeated entries become +-1 entries of a fresh column (:358-368)
    for (uint32_t i = wave; i < m; i += nwaves) {
        const uint32_t base = P.rs[i], L = len[i];
        for (uint32_t z = lane; z < L; z += 64u) {
            const uint32_t v = val[base + z], e = babs(v, p), c = col[base + z];
            if (!babsone(e, p) && (gtab_find(tab, ((uint64_t)c << rb) | e, hb) & PLO_PGCNT) >= 2u) {
                const uint32_t u = (v == e) ? 1u : p - 1u;
                col[base + z] = PLO_BFRESH; val[base + z] = u; inv[base + z] = u;
            }
        }
    }
    PLO_BIG_FENCE(); BSYNC();
    // B FactorOutRows on every row (:375-420): |v| values of the row in per-wave LDS scratch
    {
        uint32_t *sc = scratch + (size_t)wave * P.scr_stride;             // per-wave scratch, stride = longest input row
        uint32_t addacc = 0;
        for (uint32_t i = wave; i < m; i += nwaves) {
            const uint32_t base = P.rs[i], L = len[i];
            bool any = false;
            for (uint32_t z = lane; z < L; z += 64u) { uint32_t e = babs(val[base + z], p); if (babsone(e, p)) e = 0u; sc[z] = e; any |= e != 0u; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (!__ballot(any)) continue;
            uint32_t kept = 0;                                             // entries written so far (wave-uniform)
            for (uint32_t z0 = 0; z0 < L; z0 += 64u) {
                const uint32_t z = z0 + lane; const bool have = z < L;
                uint32_t c = 0, v = 0, iv = 0, e = 0, freq = 0, first = 0xFFFFFFFFu;
                if (have) { c = col[base + z]; v = val[base + z]; iv = inv[base + z]; e = sc[z]; }
                if (e != 0u) for (uint32_t y = 0; y < L; ++y) if (sc[y] == e) { ++freq; if (first == 0xFFFFFFFFu) first = y; }
                const bool grouped = e != 0u && freq > 1u, leader = grouped && first == z;
                if (leader) addacc += freq - 1u;
                const bool keep = have && (!grouped || leader);
                const uint64_t km = __ballot(keep);
                __builtin_amdgcn_wave_barrier();
                if (keep) {
                    const uint32_t np = base + kept + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                    if (leader) { col[np] = PLO_BFRESH; val[np] = e; inv[np] = (v == e) ? iv : p - iv; }
                    else { col[np] = c; val[np] = v; inv[np] = iv; }
                }
                kept += (uint32_t)__popcll(km);
            }
            if (lane == 0) len[i] = kept;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        if (addacc) wg_add(&sh.acc0, addacc);
    }
    PLO_BIG_FENCE(); BSYNC();
    if (tid == 0) { sh.nbadd += sh.acc0; sh.acc0 = 0; }
    // C Triangle (:427-507).  Columns with >= 2 non +-1 entries, ascending; rows of each in ascending order.
    for (uint32_t c = tid; c < ncols0; c += nth) tcnt[c] = 0u;
    PLO_BIG_FENCE(); BSYNC();
    for (uint32_t i = wave; i < m; i += nwaves) {
        const uint32_t base = P.rs[i], L = len[i];
        for (uint32_t z = lane; z < L; z += 64u) { const uint32_t c = col[base + z]; if (c != PLO_BFRESH && !babsone(val[base + z], p)) wg_add(&tcnt[c], 1u); }
    }
    PLO_BIG_FENCE(); BSYNC();
    if (tid == 0) {                                                        // ordered compaction (a few thousand columns)
        uint32_t nc2 = 0, off = 0;
        for (uint32_t c = 0; c < ncols0; ++c) { const uint32_t k = gload32(&tcnt[c]); if (k >= 2u) { cols2[nc2] = c; tptr2[nc2] = off; off += k; ++nc2; } }
        tptr2[nc2] = off; sh.naff = nc2;
    }
    PLO_BIG_FENCE(); BSYNC();
    const uint32_t nc2 = sh.naff;
    if (nc2) {
        // fill the row lists (binary search of the column in cols2), then sort each list
        for (uint32_t c = tid; c < ncols0; c += nth) tcnt[c] = 0u;
        PLO_BIG_FENCE(); BSYNC();
        for (uint32_t i = wave; i < m; i += nwaves) {
            const uint32_t base = P.rs[i], L = len[i];
            for (uint32_t z = lane; z < L; z += 64u) {
                const uint32_t c = col[base + z];
                if (c == PLO_BFRESH || babsone(val[base + z], p)) continue;
                uint32_t lo = 0, hi = nc2;
                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cols2[mid] < c) lo = mid + 1; else hi = mid; }
                if (lo < nc2 && cols2[lo] == c) tlist[tptr2[lo] + wg_add(&tcnt[c], 1u)] = i;
            }
        }
        PLO_BIG_FENCE(); BSYNC();
        for (uint32_t x = tid; x < nc2; x += nth) {                         // insertion sort, lists are short
            uint32_t *l = tlist + tptr2[x]; const uint32_t k = tptr2[x + 1] - tptr2[x];
            for (uint32_t u = 1; u < k; ++u) { uint32_t t = l[u], w = u; while (w > 0 && l[w - 1] > t) { l[w] = l[w - 1]; --w; } l[w] = t; }
        }
        PLO_BIG_FENCE(); BSYNC();
        if (wave == 0) {
            // A column's rows sit one per lane in PLO_TRI_R registers (round 4: 256 rows; one register = 64 rows until round 3): position
            // q = 64 r + lane, ascending q = ascending row -- the order in which the reference scans the column (:433-447).
            constexpr uint32_t R = PLO_TRI_R;
            uint32_t mulacc = 0, addacc = 0;
            for (uint32_t x = 0; x < nc2; ++x) {
                const uint32_t j = cols2[x], k = tptr2[x + 1] - tptr2[x];
                if (k > 64u * R) { if (lane == 0) wg_max(&sh.errflag, (uint32_t)BERR_PGEN); break; }
                uint32_t row[R];
#pragma unroll
                for (uint32_t r = 0; r < R; ++r) row[r] = 64u * r + lane < k ? tlist[tptr2[x] + 64u * r + lane] : 0xFFFFFFFFu;
                bool found = false;
                for (;;) {
                    uint32_t vj[R], ivj[R]; uint64_t NU[R]; uint32_t total = 0;
#pragma unroll
                    for (uint32_t r = 0; r < R; ++r) {
                        vj[r] = 0; ivj[r] = 0; bool mine = false;
                        if (row[r] != 0xFFFFFFFFu) {
                            const uint32_t base = P.rs[row[r]], L = len[row[r]];
                            for (uint32_t z = 0; z < L; ++z) if (col[base + z] == j) { vj[r] = val[base + z]; ivj[r] = inv[base + z]; mine = !babsone(vj[r], p); break; }
                        }
                        NU[r] = __ballot(mine); total += (uint32_t)__popcll(NU[r]);
                    }
                    if (total < 2u) break;
                    // first and second position holding a non +-1 entry (after the first hit only that couple is looked at again, :453-498)
                    int q1 = -1, q2 = -1;
#pragma unroll
                    for (uint32_t r = 0; r < R; ++r) { uint64_t mm = NU[r]; while (mm && q2 < 0) { const int q = (int)(64u * r) + (int)__builtin_ctzll(mm); mm &= mm - 1ull; if (q1 < 0) q1 = q; else q2 = q; } }
                    int it = -1, nx = -1;
                    for (uint32_t ri = 0; ri < R && it < 0; ++ri) {
                        uint64_t scan = NU[ri];
                        while (scan) {
                            const uint32_t li = (uint32_t)__builtin_ctzll(scan); scan &= scan - 1ull;
                            const int qi = (int)(64u * ri + li);
                            if (found && qi != q1) { scan = 0; break; }
                            uint32_t iv1 = 0;
#pragma unroll
                            for (uint32_t r = 0; r < R; ++r) if (r == ri) iv1 = (uint32_t)__shfl((int)ivj[r], (int)li);
                            int first = -1;
#pragma unroll
                            for (uint32_t r = 0; r < R; ++r) {
                                const int q = (int)(64u * r + lane);
                                bool cand = ((NU[r] >> lane) & 1ull) && q != qi;
                                if (found) cand = cand && q == (qi == q1 ? q2 : q1);
                                bool hit = false;
                                if (cand) {
                                    const uint32_t quot = bmul(vj[r], iv1, p, mu, mers), nq = p - quot;
                                    const uint32_t base = P.rs[row[r]], L = len[row[r]];
                                    for (uint32_t z = 0; z < L; ++z) {
                                        const uint32_t tv = val[base + z];
                                        if (col[base + z] != j && !babsone(tv, p) && (tv == quot || tv == nq)) { hit = true; break; }
                                    }
                                }
                                const uint64_t hm = __ballot(hit);
                                if (hm && first < 0) first = (int)(64u * r) + (int)__builtin_ctzll(hm);
                            }
                            if (first >= 0) { it = qi; nx = first; break; }
                            if (found) { scan = 0; break; }
                        }
                        if (found) break;                                   // (only the first position is a pivot then)
                    }
                    if (it < 0) break;
                    found = true;                                           // :453-498
                    uint32_t v1 = 0, iv1 = 0, vn = 0, ivn = 0;
#pragma unroll
                    for (uint32_t r = 0; r < R; ++r) {
                        if ((uint32_t)(it >> 6) == r) { v1 = (uint32_t)__shfl((int)vj[r], it & 63); iv1 = (uint32_t)__shfl((int)ivj[r], it & 63); }
                        if ((uint32_t)(nx >> 6) == r) { vn = (uint32_t)__shfl((int)vj[r], nx & 63); ivn = (uint32_t)__shfl((int)ivj[r], nx & 63); }
                    }
                    ++mulacc;
                    if (lane == 0) if (!gtab_flag(tab, ((uint64_t)j << rb) | v1, PLO_PGFLAG, hb)) wg_max(&sh.errflag, (uint32_t)BERR_TABLE);
                    uint32_t rit = 0, rnx = 0;
#pragma unroll
                    for (uint32_t r = 0; r < R; ++r) { if ((uint32_t)(it >> 6) == r) rit = (uint32_t)__shfl((int)row[r], it & 63); if ((uint32_t)(nx >> 6) == r) rnx = (uint32_t)__shfl((int)row[r], nx & 63); }
                    uint32_t addone = 0;
                    if (lane == 0) {
                        {   const uint32_t base = P.rs[rit], L = len[rit];
                            for (uint32_t z = 0; z < L; ++z) if (col[base + z] == j) { col[base + z] = PLO_BFRESH; val[base + z] = 1u; inv[base + z] = 1u; break; } }
                        const uint32_t quot = bmul(vn, iv1, p, mu, mers), iquot = bmul(ivn, v1, p, mu, mers);
                        const uint32_t eq = babs(quot, p), ieq = (quot == eq) ? iquot : p - iquot;
                        const uint32_t base = P.rs[rnx], L = len[rnx];
                        uint32_t w = 0, f = 1;
                        for (uint32_t z = 0; z < L; ++z) {
                            const uint32_t cz = col[base + z], tv = val[base + z], ti = inv[base + z];
                            if (cz == j) continue;
                            if (babs(tv, p) == eq) { ++f; continue; }
                            col[base + w] = cz; val[base + w] = tv; inv[base + w] = ti; ++w;
                        }
                        col[base + w] = PLO_BFRESH; val[base + w] = eq; inv[base + w] = ieq; ++w;
                        len[rnx] = w;
                        addone = f - 1u;
                    }
                    addacc += (uint32_t)__shfl((int)addone, 0);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __builtin_amdgcn_wave_barrier();
                }
            }
            if (lane == 0) { sh.nbmul += mulacc; sh.nbadd += addacc; }
        }
        PLO_BIG_FENCE(); BSYNC();
    }
    // D output rows (:547-604)
    {
        uint32_t addacc = 0, mulacc = 0;
        for (uint32_t i = wave; i < m; i += nwaves) {
            const uint32_t base = P.rs[i], L = len[i];
            if (lane == 0 && L > 1u) addacc += L - 1u;
            for (uint32_t z = lane; z < L; z += 64u) {
                const uint32_t e = babs(val[base + z], p), c = col[base + z];
                if (!babsone(e, p)) {
                    const bool reuse = c != PLO_BFRESH && (gtab_find(tab, ((uint64_t)c << rb) | e, hb) & PLO_PGFLAG);
                    if (!reuse) ++mulacc;
                }
            }
        }
        if (addacc) wg_add(&sh.acc0, addacc);
        if (mulacc) wg_add(&sh.acc1, mulacc);
    }
    BSYNC();
    if (sh.errflag) { if (tid == 0) atomicMax(errw, sh.errflag); return 0; }
    return ((uint64_t)(sh.nbadd + sh.acc0) << 32) | (sh.nbmul + sh.acc1);
}

// IDK: mode 2 with ratio IDENTIFIERS in the pair keys (big_candidate): moduli too wide for a residue in the 48-bit key
template <int MODE, bool DEFER, bool IDK = false> __global__ __launch_bounds__(PLO_BIG_THREADS, 4) void cse_big_kernel(BigPlan P, BigJob J)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t bigdyn[];                 // hist[maxf0+1], the value / ratio tables, then per-wave scratch (nwaves * stride)
    __shared__ BigShared sh;
    __shared__ unsigned long long cur;
    uint32_t *hist = bigdyn;
    uint32_t *nextw = bigdyn + ((P.maxf0 + 2u) & ~1u);
#ifdef PLO_BIG_DIRECT
    BigTabs TB{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#else
    BigTabs TB{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#endif
    if constexpr (MODE == 1) {                                     // {value, inverse} per value index
        uint2 *vts = (uint2 *)nextw; nextw += 2u * ((P.nv + 1u) & ~1u);
        for (uint32_t k = threadIdx.x; k < P.nv; k += blockDim.x) vts[k] = P.vt[k];
        TB.vts = vts;
    }
    if constexpr (MODE == 2) {                                     // ratio identifiers, ratios, inverse identifiers
        uint32_t *rv = nextw; nextw += (P.nr + 1u) & ~1u;
        uint16_t *rt = (uint16_t *)nextw; nextw += PLO_RSTRIDE * PLO_RSTRIDE / 2u;
        uint16_t *iv = (uint16_t *)nextw; nextw += (P.nr + 3u) / 4u * 2u;
        for (uint32_t k = threadIdx.x; k < P.nr; k += blockDim.x) { rv[k] = P.rval[k]; iv[k] = P.invid[k]; }
        for (uint32_t k = threadIdx.x; k < PLO_RSTRIDE * PLO_RSTRIDE; k += blockDim.x) rt[k] = P.rtid[k];
        TB.rval = rv; TB.rtid = rt; TB.invid = iv;
#ifdef PLO_BIG_DIRECT
        uint8_t *ng = (uint8_t *)nextw; nextw += PLO_RSTRIDE / 4u;
        if (threadIdx.x < PLO_RSTRIDE) ng[threadIdx.x] = P.negidx[threadIdx.x];
        TB.negidx = ng;
#endif
        if constexpr (!DEFER) { TB.list = (uint16_t *)nextw; nextw += (1u << P.aggbits) / 2u; }
    }
    if constexpr (DEFER) { TB.bloom = nextw; nextw += PLO_DBLOOM_WORDS; }           // Bloom filter of the hot triples; with the scratch region behind it: the merge's 64 KB
    uint32_t *scratch = nextw;                                     // ProgramGen scratch and the CSE aggregation table share this space
    uint64_t *agg = (uint64_t *)scratch;
    __syncthreads();
    uint8_t *ws = P.ws + (uint64_t)blockIdx.x * P.ws_stride;
    uint64_t best = ~0ull;
    for (;;) {
        if (threadIdx.x == 0) cur = atomicAdd(J.next, 1ull);
        __syncthreads();
        const uint64_t c = cur;
        __syncthreads();
        if (c >= J.ncand) break;
        const uint64_t seed = J.seeds ? J.seeds[c] : J.seed0 + c;
        const unsigned long long tk0 = wall_clock64();
        uint64_t ok = big_candidate<MODE, DEFER, IDK>(P, ws, seed, sh, hist, agg, P.aggbits, TB, J.err);
        uint64_t res = 0;
        __syncthreads();
        const unsigned long long tk1 = wall_clock64();
        if (ok) res = big_program_gen(P, ws, sh, scratch, J.err);
        __syncthreads();
        const unsigned long long tk2 = wall_clock64();
        if (sh.errflag) ok = 0;
        if (threadIdx.x == 0) {
            const uint32_t a = (uint32_t)(res >> 32), mu_ = (uint32_t)res;
            if (J.adds) J.adds[c] = a;
            if (J.muls) J.muls[c] = mu_;
            if (J.stats) { atomicAdd(&J.stats[32], sh.steps); atomicAdd(&J.stats[33], sh.fullscans); atomicAdd(&J.stats[34], sh.rebuilds); atomicAdd(&J.stats[35], sh.nbisect); atomicAdd(&J.stats[36], sh.spilltot); atomicAdd(&J.stats[37], sh.listover); atomicAdd(&J.stats[38], 1u); atomicAdd(&J.stats[39], sh.nforced); atomicMax(&J.stats[43], sh.derr); for (int q = 0; q < 4; ++q) { J.stats[44 + q] = (uint32_t)(sh.tmg[q] / 100ull); J.stats[48 + q] = (uint32_t)(sh.tmb[q] / 100ull); } J.stats[52] = sh.ngrp; J.stats[55] = (uint32_t)((tk1 - tk0) / 100ull); J.stats[56] = (uint32_t)((tk2 - tk1) / 100ull); atomicAdd(&J.stats[53], sh.nwin); atomicAdd(&J.stats[54], sh.nsearched); atomicAdd(&J.stats[40], sh.hotops); { const uint32_t lo_ = atomicAdd(&J.stats[41], sh.logtot_lo); if (lo_ + sh.logtot_lo < lo_) atomicAdd(&J.stats[42], 1u); atomicAdd(&J.stats[42], sh.logtot_hi); }
                J.stats[0] = sh.steps; J.stats[1] = sh.fullscans; J.stats[2] = sh.rebuilds; for (int q = 0; q < 8; ++q) J.stats[4 + q] = (uint32_t)(sh.tph[q] / 100ull);
#ifdef PLO_BIG_PROFILE
                for (int q = 0; q < 4; ++q) { J.stats[16 + q] = (uint32_t)(sh.tb1[q] / 100ull); J.stats[20 + q] = (uint32_t)(sh.tb2[q] / 100ull); J.stats[24 + q] = sh.nb[q]; }
                J.stats[28] = sh.fb1; J.stats[29] = sh.fb2; J.stats[30] = sh.fl1; J.stats[31] = sh.fl2;
                for (int q = 0; q < 16; ++q) atomicAdd(&g_prof[q], sh.pw[q]);
                for (int c_ = 0; c_ < 4; ++c_) for (int q = 0; q < 8; ++q) atomicAdd(&g_prof2[c_ * 8 + q], sh.tpc[c_][q]);
                atomicAdd(&g_prof2[32], 1ull);
#endif
            }   // phase times in us
            // 64-bit cost word: the op-counts of config 5 do not fit 16 bits
            uint64_t ck;
            switch (J.cost_mode) { case 1: ck = ((uint64_t)a << 20) | mu_; break; case 2: ck = (uint64_t)(a + mu_) << 20; break; default: ck = ((uint64_t)(a + mu_) << 20) | a; }
            const uint64_t packed = (ck << 24) | (c & 0xFFFFFFull);
            if (ok) best = packed < best ? packed : best;
        }
        __syncthreads();
    }
    if (J.best && threadIdx.x == 0 && best != ~0ull) atomicMin(J.best, (unsigned long long)best);
}

} // namespace plo
