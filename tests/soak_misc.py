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
            G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)])
        else:
            G = TrilPlan(O.m, [(n_, rp, col, [int(x) for x in num], [int(x) for x in den]) for n_, (rp, col, num, den) in zip(O.dims, O.csr)])
        got = G.cost_many(seed0=s, n=12)
    except capi.PloError as e:
        if e.code in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
            refused2 += 1
            continue
        raise
    ran2 += 1
    if got != O.cost_many(seed0=s, nseeds=12):
        bad += 1
        print("MISMATCH trilinear case", s - 1, m, na, nb, nc, unit, flush=True)
print("# trilinear: %d triples, %d refused; total mismatches %d in %.0f s" % (ran2, refused2, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
