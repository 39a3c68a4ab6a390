#!/usr/bin/env python3
"""One-off soak of the HBM-resident kernel family against the CPU oracle (run on the GPU box: python tests/soak_hbm.py [seconds]):
every matrix of the reference's data/ forced through plo_cse_big.hip (PLO_PLAN_HBM), and random matrices whose steps touch more than
2048 entries per 64-row batch of a wave (several windows of the flat sweep), a few seeds each, bit-exact (adds, muls) per seed.
Not collected by pytest; prints one line per failure and a summary."""
import glob
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth                                                    # noqa: E402
from plo_testlib import DATA, OracleMatrix                      # noqa: E402
from plinopt_amd import CSEPlan, capi                           # noqa: E402

P = 131071
if os.environ.get("PLO_SOAK_TRACE"):
    import faulthandler
    _tf = open(os.environ["PLO_SOAK_TRACE"], "w")
    faulthandler.enable(file=_tf)
    faulthandler.dump_traceback_later(45, repeat=True, file=_tf)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
t0 = time.time()
bad = ran = 0


def check(tag, M, seed0, n):
    global bad, ran
    plan = CSEPlan(M.m, M.n, M.rowptr, M.col, M.val, M.p, hbm=True)
    got = plan.cost_many(seed0=seed0, n=n)
    plan.close()
    exp = tuple(M.cost_many(seed0=seed0, nseeds=n, nthreads=8))
    ran += 1
    if got != exp:
        bad += 1
        print("MISMATCH", tag, seed0, [k for k in range(n) if (got[0][k], got[1][k]) != (exp[0][k], exp[1][k])][:5], flush=True)


for f in sorted(glob.glob(os.path.join(DATA, "*.sms"))):
    if time.time() - t0 > budget * 0.5:
        break
    try:
        M = OracleMatrix.from_sms(f, P)
    except Exception:
        continue                                                # the -X files (symbolic entries)
    if M.m * M.n > 20000:
        continue
    check(os.path.basename(f), M, 1000, 12)
print("# data matrices:", ran, "checked,", bad, "mismatches", flush=True)
if os.environ.get("PLO_SOAK_WIDE"):
    # moduli of 25 to 31 bits, at most 32 distinct coefficients, up to 2000 columns: residues in the 48-bit pair keys while they leave room
    # for the columns, ratio identifiers beyond (plo::cse_big_kernel<2, ., true>); against the literal oracle
    s = 0; wide = refused = 0
    while time.time() - t0 < budget:
        rng = random.Random(7700 + s); s += 1
        p = rng.choice([2147483629, 2147483647, 1073741827, 16777259, 536870923])
        m, n = rng.randint(10, 120), rng.randint(8, 2000)
        per_row = rng.randint(2, 14)
        vals = [1, p - 1] + [rng.randint(2, p - 2) for _ in range(rng.choice([0, 1, 3, 12, 30]))]
        rows = [{j: rng.choice(vals) for j in rng.sample(range(n), min(n, per_row))} for _ in range(m)]
        rp, c, v = synth.to_csr(rows, p)
        try:
            check("wide %d %dx%d" % (p, m, n), OracleMatrix(m, n, rp, c, v, p), s * 5, 4)
            wide += 1
        except capi.PloError as e:
            if e.code not in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):
                raise
            refused += 1
            print("refused:", str(e)[:140], flush=True)
    print("# wide moduli: %d matrices, %d refused; total %d checked, %d mismatches in %.0f s" % (wide, refused, ran, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)
import re              # noqa: E402
import subprocess      # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPT = os.path.join(ROOT, "bin", "optimizer")
s = 0
while time.time() - t0 < budget:
    # larger random matrices (steps of several hundred rows of ~100 entries: several 2048-entry windows per wave batch) against the
    # build's scalable host engine (`bin/optimizer --replay`, which prints the oracle's text wherever the oracle can walk)
    rng = random.Random(7000 + s)
    m, n = rng.randint(150, 400), rng.randint(150, 320)
    dens = rng.choice([0.2, 0.35, 0.5])
    vals = [1, P - 1] + [rng.randint(2, P - 2) for _ in range(rng.randint(0, 3))]
    if os.environ.get("PLO_SOAK_MANY_VALUES") and s % 2:        # more than 32 / more than 512 distinct values: kernel modes 1 and 0 (value table in LDS / in global memory)
        m, n, dens = rng.randint(100, 200), rng.randint(100, 200), rng.choice([0.2, 0.35])
        vals = [1, P - 1] + [rng.randint(2, P - 2) for _ in range(rng.choice([40, 700]))]
    if os.environ.get("PLO_SOAK_BIG"):                          # thousands of rows: steps of more than 1000 rows, logs that force merges
        m, n, dens = rng.randint(500, 2500), rng.randint(200, 800), rng.choice([0.04, 0.08, 0.15])
    rows = [{j: rng.choice(vals) for j in range(n) if rng.random() < dens} for _ in range(m)]
    rows = [r if r else {0: 1} for r in rows]
    rp, c, v = synth.to_csr(rows, P)
    path = "/tmp/plo_soak_%d.sms" % os.getpid()
    with open(path, "w") as f:
        f.write("%d %d M\n" % (m, n))
        for i, r in enumerate(rows):
            for j in sorted(r):
                f.write("%d %d %d\n" % (i + 1, j + 1, r[j] if r[j] <= P // 2 else r[j] - P))
        f.write("0 0 0\n")
    if os.environ.get("PLO_SOAK_TRACE"):
        print("case", s, m, n, dens, len(vals), flush=True)
    plan = CSEPlan(m, n, rp, c, v, P, hbm=True)
    try:
        got = plan.cost_many(seed0=s * 10, n=3)
    except Exception as e:
        plan.close()
        if getattr(e, "code", 0) in (capi.PLO_E_CAPACITY, capi.PLO_E_UNSUPPORTED):      # a stated limit (the tools then search on the host)
            print("refused:", str(e)[:140], flush=True)
            s += 1
            continue
        raise
    plan.close()
    for k in range(3):
        r = subprocess.run([OPT, "-q", str(P), "--gpu", "0", "--replay", "--seed", str(s * 10 + k), path], capture_output=True, text=True)
        g1, g2 = re.search(r"# (\d+)\tadditions", r.stderr), re.search(r"# (\d+)\tmultiplications", r.stderr)
        ran += 1
        if not (g1 and g2) or (got[0][k], got[1][k]) != (int(g1.group(1)), int(g2.group(1))):
            bad += 1
            print("MISMATCH random %dx%d dens %.2f values %d seed %d: GPU %s host %s" % (m, n, dens, len(vals), s * 10 + k, (got[0][k], got[1][k]), (g1 and g1.group(1), g2 and g2.group(1))), flush=True)
    s += 1
print("# total:", ran, "checked,", bad, "mismatches in %.0f s" % (time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
