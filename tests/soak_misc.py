#!/usr/bin/env python3
"""One-off soak of the kernel-method and in-place trilinear kernels against the CPU oracle on random inputs (GPU box:
python tests/soak_misc.py [seconds]).  Refusals (PLO_E_UNSUPPORTED / PLO_E_CAPACITY) are counted, anything else is an error."""
import os
import random
import sys
import time
from fractions import Fraction

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth                                                    # noqa: E402
from plo_testlib import OracleMatrix, OracleTril                # noqa: E402
from plinopt_amd import TrilPlan, capi, kernel_search           # noqa: E402

P = 131071
capi.check(capi.lib().plo_init(0))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0 = time.time()
bad = ran = refused = 0
s = 0
while time.time() - t0 < budget / 2:                            # ---- kernel method
    rng = random.Random(11000 + s); s += 1
    n = rng.randint(2, 24); m = rng.randint(n + 1, min(128, n + 40))
    vals = [1, P - 1] if rng.random() < 0.7 else [1, P - 1, 2, P - 2, 3]
    rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < rng.choice([0.2, 0.5])} for _ in range(m)]
    rows = [r if r else {rng.randrange(n): 1} for r in rows]
    rp, c, v = synth.to_csr(rows, P)
    M = OracleMatrix(m, n, rp, c, v, P)
    if M.kernel_restart(1) is None:
        continue
    try:
        adds, muls, info, best, st = kernel_search((m, n, rp, c, v), P, s, 12)
    except capi.PloError as e:
        if e.code in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
            refused += 1
            continue
        raise
    ran += 1
    exp = [M.kernel_restart(s + k) for k in range(12)]
    if [(a, mu) + i for a, mu, i in zip(adds, muls, info)] != exp:
        bad += 1
        print("MISMATCH kernel method case", s - 1, m, n, flush=True)
print("# kernel method: %d matrices, %d refused, %d mismatches" % (ran, refused, bad), flush=True)
ran2 = refused2 = 0
vals_q = [Fraction(1), Fraction(-1), Fraction(1, 2), Fraction(-2), Fraction(3), Fraction(-2, 3)]
while time.time() - t0 < budget:                                # ---- in-place trilinear search
    rng = random.Random(12000 + s); s += 1
    m = rng.randint(2, 40); na, nb, nc = rng.randint(2, 12), rng.randint(2, 12), rng.randint(2, 12)
    unit = rng.random() < 0.6
    expanded = rng.random() < 0.35                              # `trilplacer -e` (round 4: rational inputs too)
    def mat(rows, cols):
        e = {}
        for i in range(rows):
            js = [j for j in range(cols) if rng.random() < 0.4] or [rng.randrange(cols)]
            for j in js:
                e[(i, j)] = rng.choice(vals_q[:2] if unit else vals_q)
        return e
    A, B = (m, na, mat(m, na)), (m, nb, mat(m, nb))
    T = mat(m, nc)                                              # T = C^T: m x nc, every row non-empty
    C = (nc, m, {(j, i): v for (i, j), v in T.items()})
    O = OracleTril(A, B, C)
    try:
        if unit:
            G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=expanded)
        else:
            G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num], [int(x) for x in den]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)], expanded=expanded)
        got = G.cost_many(seed0=s, n=12)
    except capi.PloError as e:
        if e.code in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
            refused2 += 1
            continue
        raise
    ran2 += 1
    if got != O.cost_many(seed0=s, nseeds=12, expanded=expanded):
        bad += 1
        print("MISMATCH trilinear case", s - 1, m, na, nb, nc, unit, expanded, flush=True)
print("# trilinear: %d triples, %d refused; total mismatches %d in %.0f s" % (ran2, refused2, bad, time.time() - t0), flush=True)
if "--cob" in sys.argv:                                         # ---- change-of-basis enumeration (and its batch entry)
    from plo_testlib import oracle_cob_search
    from plinopt_amd import cob_search, cob_search_batch
    ran3 = 0
    t1 = time.time()
    while time.time() - t1 < budget / 2:
        rng = random.Random(13000 + s); s += 1
        n, m = rng.randint(1, 13), rng.randint(1, 70)
        row = rng.randint(0, n - 1); off = (row // 4) * 4
        probs = []
        for _ in range(rng.randint(1, 4)):
            p = rng.choice([7, 101, 131071, 2147483629, 2147483647])
            TM = [rng.choice([0, 0, 1, p - 1, 2, 3, 5]) % p for _ in range(n * m)]
            Cand = [0] * (n * n)
            for i in range(row):
                for j in range(n):
                    Cand[i * n + j] = rng.choice([0, 0, 1, p - 1, 2]) % p
            C = rng.randint(1, 9)
            coeffs = [0, 1, p - 1, 2 % p, (p - 2) % p, 3 % p, pow(2, -1, p), (p - pow(2, -1, p)) % p, 5 % p][:C]
            w0 = rng.choice([-1, 0, m // 2, m])
            probs.append((TM, Cand, coeffs, p, w0, 0 if w0 >= 0 else -1))
        exp = [oracle_cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, w1) for (TM, Cand, coeffs, p, w0, w1) in probs]
        got, _ = cob_search_batch(n, m, row, off, probs)
        one = [cob_search(n, m, TM, Cand, row, off, coeffs, p, w0, w1)[0] for (TM, Cand, coeffs, p, w0, w1) in probs]
        ran3 += 1
        if got != exp or one != exp:
            bad += 1
            print("MISMATCH CoB case", s - 1, n, m, row, flush=True)
    print("# CoB: %d groups of 1-4 enumerations (single and batched launches), total mismatches %d" % (ran3, bad), flush=True)
if "--enum" in sys.argv:                                        # ---- schedule enumeration (-E) and the chained candidates of -G
    from plinopt_amd import CSEPlan, CSEChain
    ran4 = ran5 = 0
    t1 = time.time()
    while time.time() - t1 < budget / 2:
        rng = random.Random(14000 + s); s += 1
        p = rng.choice([7, 131071, 2147483629])
        m, n = rng.randint(2, 14), rng.randint(2, 12)
        vals = [1, p - 1] if rng.random() < 0.5 else [1, p - 1, 2 % p or 1, 3 % p or 1]
        rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < 0.5} for _ in range(m)]
        rows = [r if r else {0: 1} for r in rows]
        rp, c, v = synth.to_csr(rows, p)
        M = OracleMatrix(m, n, rp, c, v, p)
        try:
            plan = CSEPlan(m, n, rp, c, v, p)
            first = rng.choice([0, 0, 17, 1000])
            got = plan.enum_cost_many(first, 96)
            plan.close()
        except capi.PloError as e:
            if e.code in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
                continue
            raise
        ran4 += 1
        if got != tuple(M.enum_cost_many(first, 96, nthreads=8)):
            bad += 1
            print("MISMATCH enumeration case", s - 1, m, n, p, flush=True)
    print("# schedule enumeration: %d matrices x 96 schedules, total mismatches %d" % (ran4, bad), flush=True)
sys.exit(1 if bad else 0)
