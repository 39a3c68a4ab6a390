#!/usr/bin/env python3
"""One-off soak of the LDS-resident wave kernel against the CPU oracle (run on the GPU box: python tests/soak_wave.py [seconds]):
random matrices of up to 64 x 64 (rows of at most 64 entries; +-1 only with up to 200 rows) over several moduli, 8 seeds each,
bit-exact (adds, muls) per seed; a capacity refusal of the wave kernel (-> the HBM family) is counted, not an error."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth                                                    # noqa: E402
from plo_testlib import OracleMatrix                            # noqa: E402
from plinopt_amd import CSEPlan, capi                           # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t0 = time.time()
bad = ran = hbm = 0
s = 0
while time.time() - t0 < budget:
    rng = random.Random(9000 + s)
    p = rng.choice([7, 101, 131071, 2147483629, 2147483647])
    unit = rng.random() < 0.4
    m = rng.randint(2, 200 if unit else 64)
    n = rng.randint(2, 64)
    dens = rng.choice([0.1, 0.3, 0.6, 0.9])
    vals = [1, p - 1] if unit else [1, p - 1] + [rng.randint(2, p - 2) % p or 1 for _ in range(rng.choice([1, 3, 30]))]
    rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < dens} for _ in range(m)]
    rows = [r if r else {0: 1} for r in rows]
    if sum(len(r) for r in rows) > 700:                         # (the literal oracle rescans the pair map at every step)
        s += 1
        continue
    rp, c, v = synth.to_csr(rows, p)
    M = OracleMatrix(m, n, rp, c, v, p)
    try:
        plan = CSEPlan(m, n, rp, c, v, p)
        hbm += 1 if plan.is_hbm else 0
        got = plan.cost_many(seed0=s, n=8)
        plan.close()
    except Exception as e:
        if getattr(e, "code", 0) in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
            print("refused:", m, n, dens, len(vals), p, str(e)[:100], flush=True)
            s += 1
            continue
        raise
    exp = tuple(M.cost_many(seed0=s, nseeds=8, nthreads=8))
    ran += 1
    if got != exp:
        bad += 1
        print("MISMATCH case", s, m, n, dens, len(vals), p, flush=True)
    s += 1
print("# %d matrices (%d on the HBM family), %d mismatches in %.0f s" % (ran, hbm, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
