"""Host-side mirror of the reference's restart-loop interface for the CSE path.

`CSEOptimiser(nbops, sout, F, M, T, global, randomloops, verbose)`
(reference include/plinopt_optimize.inl:1193-1247) runs `randomloops`
independent candidates and keeps the best (adds, muls) under `cmpOpCount`
(include/plinopt_optimize.h:53-64).  Here the loop body runs on the GPU through
libplinopt_hip.so; this module only marshals arguments.
"""
import ctypes

from . import capi


class CSEPlan:
    """A matrix over Z_p prepared once and resident in HBM (plo_cse_plan_create)."""

    def __init__(self, m, n, rowptr, col, val, p, device=None, hbm=False):
        L = capi.lib()
        if device is not None:
            capi.check(L.plo_init(int(device)))
        self.m, self.n, self.p = m, n, p
        self.nnz = len(col)
        csr, self._keep = capi.make_csr(m, n, rowptr, col, val)
        h = ctypes.c_void_p()
        capi.check(L.plo_cse_plan_create_ex(ctypes.byref(csr), p, capi.PLAN_HBM if hbm else 0, ctypes.byref(h)))
        self._h = h
        self.is_hbm = bool(L.plo_cse_plan_is_hbm(h))     # HBM-resident kernel family (one workgroup per candidate)
        self.last_stats = None

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().plo_cse_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def hbm_counters(self):
        """Diagnostics of the HBM-resident family over the last launch (plo_cse_plan_hbm_counters)."""
        out = (ctypes.c_uint32 * 10)()
        capi.check(capi.lib().plo_cse_plan_hbm_counters_ex(self._h, out, 10))
        return dict(zip(("steps", "full_scans", "level_rebuilds", "bisections", "spilled_pairs", "list_overflows", "candidates", "eager_refits", "extra_sweep_windows", "rows_searched"), out[:10]))

    def cost_many(self, seeds=None, seed0=0, n=0):
        """(adds[], muls[]) of candidates `seeds` (or seed0..seed0+n-1): Optimizer() per seed."""
        L = capi.lib()
        if seeds is not None:
            n = len(seeds)
            sp = (ctypes.c_uint64 * max(n, 1))(*seeds)
        else:
            sp = None
        adds = (ctypes.c_uint32 * max(n, 1))()
        muls = (ctypes.c_uint32 * max(n, 1))()
        st = capi.Stats()
        capi.check(L.plo_cse_cost_many_plan(self._h, sp, seed0, n, adds, muls, ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return list(adds[:n]), list(muls[:n])

    def search(self, seed0, nseeds, cost_mode=capi.COST_SUM_THEN_ADD):
        """Best (adds, muls, seed) over seeds seed0..seed0+nseeds-1, ties to the smallest seed."""
        L = capi.lib()
        b, st = capi.Best(), capi.Stats()
        capi.check(L.plo_cse_search_plan(self._h, seed0, nseeds, cost_mode, ctypes.byref(b), ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return b.adds, b.muls, b.seed

    # -E: RecSub's schedule tree walked by index (include/plinopt_hip.h)
    def enum_cost_many(self, first, n):
        """(adds, muls, product of radices) of the schedules first..first+n-1"""
        L = capi.lib()
        adds = (ctypes.c_uint32 * max(n, 1))(); muls = (ctypes.c_uint32 * max(n, 1))(); prods = (ctypes.c_uint64 * max(n, 1))()
        st = capi.Stats()
        capi.check(L.plo_cse_enum_cost_many_plan(self._h, first, n, adds, muls, prods, ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return list(adds[:n]), list(muls[:n]), list(prods[:n])

    def enum_search(self, first, count, cost_mode=capi.COST_ADD_THEN_MUL):
        """Best (adds, muls, index) over the schedules first..first+count-1 and the largest radix product seen: the
        enumeration from 0 is exhaustive once count >= that product."""
        L = capi.lib()
        b, st, mp = capi.Best(), capi.Stats(), ctypes.c_uint64()
        capi.check(L.plo_cse_enum_search_plan(self._h, first, count, cost_mode, ctypes.byref(b), ctypes.byref(mp), ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return (b.adds, b.muls, b.seed), mp.value


def cse_search_multi(m, n, rowptr, col, val, p, seed0, nseeds, devices, cost_mode=capi.COST_SUM_THEN_ADD):
    """One process, several devices (plo_cse_search_multi): the restart range in len(devices) contiguous shards, one host thread and
    one device each; (adds, muls, seed) of the minimum under (cmpOpCount key, seed) and the aggregated stats."""
    L = capi.lib()
    csr, keep = capi.make_csr(m, n, rowptr, col, val)
    devs = (ctypes.c_int * len(devices))(*devices)
    b, st = capi.Best(), capi.Stats()
    capi.check(L.plo_cse_search_multi(ctypes.byref(csr), p, seed0, nseeds, cost_mode, len(devices), devs, ctypes.byref(b), ctypes.byref(st)))
    del keep
    return (b.adds, b.muls, b.seed), {"seconds": st.seconds, "kernel_ms": st.kernel_ms, "candidates": st.candidates, "launches": st.launches, "reduce": st.reduce}


def cmp_op_count_key(adds, muls, cost_mode=capi.COST_SUM_THEN_ADD):
    """Sort key equivalent to cmpOpCount (include/plinopt_optimize.h:53-64)."""
    if cost_mode == capi.COST_ADD_THEN_MUL:
        return (adds, muls)
    if cost_mode == capi.COST_SUM:
        return (adds + muls,)
    return (adds + muls, adds)


class CSEChain:
    """Two matrices per candidate with one random stream: the restart loop of `LUOptimiser`
    (reference include/plinopt_optimize.inl:1056-1100) -- Optimizer() on U, then on L, costs added."""

    def __init__(self, first, second, p, device=None):
        L = capi.lib()
        if device is not None:
            capi.check(L.plo_init(int(device)))
        c1, self._k1 = capi.make_csr(*first)
        c2, self._k2 = capi.make_csr(*second)
        h = ctypes.c_void_p()
        capi.check(L.plo_cse_chain_create(ctypes.byref(c1), ctypes.byref(c2), p, ctypes.byref(h)))
        self._h = h
        self.last_stats = None

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().plo_cse_chain_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cost_many(self, seeds=None, seed0=0, n=0):
        L = capi.lib()
        if seeds is not None:
            n = len(seeds)
            sp = (ctypes.c_uint64 * max(n, 1))(*seeds)
        else:
            sp = None
        adds = (ctypes.c_uint32 * max(n, 1))()
        muls = (ctypes.c_uint32 * max(n, 1))()
        st = capi.Stats()
        capi.check(L.plo_cse_chain_cost_many(self._h, sp, seed0, n, adds, muls, ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return list(adds[:n]), list(muls[:n])

    def search(self, seed0, nseeds, cost_mode=capi.COST_SUM_THEN_ADD):
        L = capi.lib()
        b, st = capi.Best(), capi.Stats()
        capi.check(L.plo_cse_chain_search(self._h, seed0, nseeds, cost_mode, ctypes.byref(b), ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return b.adds, b.muls, b.seed


def chain_batch(pairs, p, seed0, per_pair, cost_mode=capi.COST_SUM_THEN_ADD, want_costs=True):
    """Many chained pairs in one launch (`plo_cse_chain_batch`): pairs = [((m,n,rowptr,col,val), (m,n,rowptr,col,val)), ...];
    candidate c runs on pair c // per_pair with seed seed0 + c.  Returns (adds, muls, (best adds, best muls, best seed), stats)."""
    L = capi.lib()
    keep, firsts, seconds = [], (capi.CSR * len(pairs))(), (capi.CSR * len(pairs))()
    for k, (A, B) in enumerate(pairs):
        for dst, (m, n, rp, col, val) in ((firsts, A), (seconds, B)):
            a = ((ctypes.c_uint32 * len(rp))(*rp), (ctypes.c_uint32 * max(len(col), 1))(*col), (ctypes.c_uint32 * max(len(val), 1))(*val))
            keep.append(a)
            dst[k] = capi.CSR(m, n, a[0], a[1], a[2])
    n = len(pairs) * per_pair
    adds = (ctypes.c_uint32 * n)() if want_costs else None
    muls = (ctypes.c_uint32 * n)() if want_costs else None
    b, st = capi.Best(), capi.Stats()
    capi.check(L.plo_cse_chain_batch(len(pairs), firsts, seconds, p, seed0, per_pair, cost_mode, adds, muls, ctypes.byref(b), ctypes.byref(st)))
    return (list(adds) if want_costs else None), (list(muls) if want_costs else None), (b.adds, b.muls, b.seed), st.as_dict()


def kernel_search(M, p, seed0, nrestarts, per_block=1, cost_mode=capi.COST_SUM_THEN_ADD, want_costs=True):
    """The kernel method with decomposition, images and both Optimizer calls on the device (`plo_kernel_search`):
    M = (m, n, rowptr, col, val).  Returns (adds, muls, info, (best adds, best muls, best seed), stats); info[c] =
    (rank, NotIndep, dependent rows) of restart c."""
    L = capi.lib()
    m, n, rp, col, val = M
    A, keep = capi.make_csr(m, n, rp, col, val)
    adds = (ctypes.c_uint32 * nrestarts)() if want_costs else None
    muls = (ctypes.c_uint32 * nrestarts)() if want_costs else None
    info = (ctypes.c_uint32 * (3 * nrestarts))() if want_costs else None
    b, st = capi.Best(), capi.Stats()
    capi.check(L.plo_kernel_search(ctypes.byref(A), p, seed0, nrestarts, per_block, cost_mode, adds, muls, info, ctypes.byref(b), ctypes.byref(st)))
    del keep
    inf = [tuple(info[3 * k:3 * k + 3]) for k in range(nrestarts)] if want_costs else None
    return (list(adds) if want_costs else None), (list(muls) if want_costs else None), inf, (b.adds, b.muls, b.seed), st.as_dict()


def kernel_search_multi(M, p, seed0, nrestarts, devices, cost_mode=capi.COST_SUM_THEN_ADD):
    """`plo_kernel_search_multi`: the restart loop of the kernel method over the listed devices from this process (one host thread
    and one device per contiguous shard, minimum by the RCCL MIN all-reduces).  Returns ((adds, muls, seed), stats)."""
    L = capi.lib()
    m, n, rp, col, val = M
    A, keep = capi.make_csr(m, n, rp, col, val)
    dv = (ctypes.c_int * len(devices))(*devices)
    b, st = capi.Best(), capi.Stats()
    capi.check(L.plo_kernel_search_multi(ctypes.byref(A), p, seed0, nrestarts, 1, cost_mode, len(devices), dv, ctypes.byref(b), ctypes.byref(st)))
    del keep
    return (b.adds, b.muls, b.seed), st.as_dict()


def tril_search_multi(m, mats, seed0, nseeds, devices, expanded=False):
    """`plo_tril_search_multi` (BASELINE configs[3]): the restart loop of SearchTriLinearAlgorithm over the listed devices from this
    process.  mats = three (n, rowptr, col, num[, den]) triples as for TrilPlan.  Returns (((ADD, SCA, MUL), seed, variant), stats)."""
    L = capi.lib()
    keep, cs = [], []
    for t in mats:
        n, rowptr, col, val = t[:4]
        den = t[4] if len(t) == 5 else [1] * len(val)
        a = ((ctypes.c_uint32 * len(rowptr))(*rowptr), (ctypes.c_uint32 * max(len(col), 1))(*col), (ctypes.c_int64 * max(len(val), 1))(*[int(x) for x in val]),
             (ctypes.c_int64 * max(len(den), 1))(*[int(x) for x in den]))
        keep.append(a)
        cs.append(capi.QCSR(m, n, a[0], a[1], a[2], a[3]))
    dv = (ctypes.c_int * len(devices))(*devices)
    b, st = capi.TrilBest(), capi.Stats()
    capi.check(L.plo_tril_search_multi(ctypes.byref(cs[0]), ctypes.byref(cs[1]), ctypes.byref(cs[2]), int(bool(expanded)), seed0, nseeds, len(devices), dv,
                                       ctypes.byref(b), ctypes.byref(st)))
    del keep
    return ((b.add, b.sca, b.mul), b.seed, b.variant), st.as_dict()


def cob_search(n, m, TM, Cand, row, offsetblock, coeffs, p, w0=-1, w1=-1, groups=None):
    """One (block,row) enumeration of `localSparsifier` (reference include/plinopt_sparsify.inl:282-314) on the
    GPU: |coeffs|^4 candidate rows through `testLinComb`.  TM (n x m) and Cand (n x n) are flat row-major lists of
    residues.  groups = (first, count): only the (i,j,k) prefixes first..first+count-1 (a shard, plo_cob_search_range).
    Returns ((zeros_v, zeros_w, index, found), stats)."""
    L = capi.lib()
    b, st = capi.CobBest(), capi.Stats()
    arr = lambda xs: (ctypes.c_uint32 * max(len(xs), 1))(*xs)
    g0, gn = groups if groups is not None else (0, len(coeffs) ** 3)
    capi.check(L.plo_cob_search_range(n, m, arr(TM), arr(Cand), row, offsetblock, arr(coeffs), len(coeffs), p, w0, w1, g0, gn,
                                      ctypes.byref(b), ctypes.byref(st)))
    return (b.zeros_v, b.zeros_w, b.index, b.found), st.as_dict()


def cob_search_batch(n, m, row, offsetblock, problems):
    """Up to 4 enumerations of the same shape in ONE launch (`plo_cob_search_batch`): problems = [(TM, Cand, coeffs, p, w0, w1), ...]
    (the two primes of an enumeration over the rationals).  Returns ([(zeros_v, zeros_w, index, found), ...], stats)."""
    L = capi.lib()
    arr = lambda xs: (ctypes.c_uint32 * max(len(xs), 1))(*xs)
    keep, pr = [], (capi.CobProblem * len(problems))()
    for k, (TM, Cand, coeffs, p, w0, w1) in enumerate(problems):
        a, b, c = arr(TM), arr(Cand), arr(coeffs); keep += [a, b, c]
        pr[k] = capi.CobProblem(ctypes.cast(a, ctypes.POINTER(ctypes.c_uint32)), ctypes.cast(b, ctypes.POINTER(ctypes.c_uint32)),
                                ctypes.cast(c, ctypes.POINTER(ctypes.c_uint32)), len(coeffs), p, w0, w1)
    out, st = (capi.CobBest * len(problems))(), capi.Stats()
    capi.check(L.plo_cob_search_batch(len(problems), n, m, row, offsetblock, pr, out, ctypes.byref(st)))
    return [(b.zeros_v, b.zeros_w, b.index, b.found) for b in out], st.as_dict()


TRIL_BASE_SEED = (1 << 64) - 1


class TrilPlan:
    """Mirror of the restart loop of `SearchTriLinearAlgorithm` (reference include/plinopt_inplace.inl:812-929):
    A, B and T (= transpose of the product matrix) as integer CSR triples (rowptr, col, val) with m rows each.
    `cost_many` returns, per seed, ((ADD,SCA,MUL) oriented, (ADD,SCA,MUL) unoriented); `search` the best
    (ADD, SCA, MUL, seed, variant) under the reference's order (:893-897), ties to the smaller (seed, variant)."""

    def __init__(self, m, mats, device=None, expanded=False):
        """expanded: `trilplacer -e` (TransposedDoubleAlgorithm on the double expansion of T)"""
        L = capi.lib()
        if device is not None:
            capi.check(L.plo_init(device))
        self._keep = []
        cs = []
        rational = any(len(t) == 5 for t in mats)            # (n, rowptr, col, num, den): rational coefficients (plo_tril_plan_create_q)
        for t in mats:
            n, rowptr, col, val = t[:4]
            if rational:
                den = t[4] if len(t) == 5 else [1] * len(val)
                a = ((ctypes.c_uint32 * len(rowptr))(*rowptr), (ctypes.c_uint32 * max(len(col), 1))(*col), (ctypes.c_int64 * max(len(val), 1))(*[int(x) for x in val]),
                     (ctypes.c_int64 * max(len(den), 1))(*[int(x) for x in den]))
                cs.append(capi.QCSR(m, n, a[0], a[1], a[2], a[3]))
            else:
                a = ((ctypes.c_uint32 * len(rowptr))(*rowptr), (ctypes.c_uint32 * max(len(col), 1))(*col), (ctypes.c_int32 * max(len(val), 1))(*val))
                cs.append(capi.ICSR(m, n, a[0], a[1], a[2]))
            self._keep.append(a)
        self._h = ctypes.c_void_p()
        create = L.plo_tril_plan_create_q if rational else L.plo_tril_plan_create_x
        capi.check(create(ctypes.byref(cs[0]), ctypes.byref(cs[1]), ctypes.byref(cs[2]), int(bool(expanded)), ctypes.byref(self._h)))
        self.last_stats = None

    def __del__(self):
        try:
            if self._h:
                capi.lib().plo_tril_plan_destroy(self._h); self._h = None
        except Exception:
            pass

    def cost_many(self, seeds=None, seed0=0, n=0):
        L = capi.lib()
        if seeds is not None:
            n = len(seeds); sp = (ctypes.c_uint64 * max(n, 1))(*seeds)
        else:
            sp = None
        ops = (ctypes.c_uint32 * (6 * max(n, 1)))()
        st = capi.Stats()
        capi.check(L.plo_tril_cost_many(self._h, sp, seed0, n, ops, ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return [(tuple(ops[6 * k:6 * k + 3]), tuple(ops[6 * k + 3:6 * k + 6])) for k in range(n)]

    def search(self, seed0, nseeds):
        b, st = capi.TrilBest(), capi.Stats()
        capi.check(capi.lib().plo_tril_search(self._h, seed0, nseeds, ctypes.byref(b), ctypes.byref(st)))
        self.last_stats = st.as_dict()
        return (b.add, b.sca, b.mul), b.seed, b.variant
