#!/usr/bin/env python3
"""Config 5 driver (GPU box): regenerate 32x32x32_15096_L from its stored SLP, run candidates on the
HBM-resident kernel family, compare with the host engine's known costs.
usage: python tests/run_config5.py [ncand] [slices]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plo_testlib import DATA  # noqa: E402

P = 131071
EXPECT = {1: (34409, 4546), 2: (34402, 4544)}       # bin/optimizer --replay --engine fast (host), verified by SLPchecker


def load_l32():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
    t0 = time.time()
    out = subprocess.run([os.path.join(ROOT, "bin", "SLPchecker"), "-q", str(P), os.path.join(DATA, "32x32x32_15096_L.slp")],
                         capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    m, n = int(lines[0].split()[0]), int(lines[0].split()[1])
    rows = [[] for _ in range(m)]
    for ln in lines[1:-1]:
        i, j, v = ln.split()
        rows[int(i) - 1].append((int(j) - 1, int(v)))
    rp, c, v = [0], [], []
    for r in rows:
        r.sort()
        for j, x in r:
            c.append(j); v.append(x)
        rp.append(len(c))
    print("matrix %dx%d nnz %d regenerated in %.1f s" % (m, n, len(c), time.time() - t0), flush=True)
    return m, n, rp, c, v


def main():
    ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    from plinopt_amd import CSEPlan, capi
    capi.check(capi.lib().plo_init(0))
    m, n, rp, c, v = load_l32()
    t0 = time.time()
    plan = CSEPlan(m, n, rp, c, v, P)
    print("plan (hbm=%s) built in %.1f s" % (plan.is_hbm, time.time() - t0), flush=True)
    t0 = time.time()
    a, mu = plan.cost_many(seed0=1, n=ncand)
    dt = time.time() - t0
    st = plan.last_stats
    print("candidates %d in %.2f s (kernel %.1f ms): %.3f candidates/s; grid %d" % (ncand, dt, st["kernel_ms"], ncand / (st["kernel_ms"] * 1e-3), st["grid"]), flush=True)
    print("costs", list(zip(a, mu))[:8])
    for s, exp in EXPECT.items():
        if s - 1 < ncand:
            got = (a[s - 1], mu[s - 1])
            print("seed", s, "got", got, "expected", exp, "OK" if got == exp else "MISMATCH")
            assert got == exp


if __name__ == "__main__":
    main()
