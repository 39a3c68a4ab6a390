#!/usr/bin/env python3
"""Config 5 on the GPU box: throughput of the HBM kernel family against the number of candidates in flight
(PLO_BIG_SLICES), with the phase clocks of the last candidate.  usage: python tests/sweep_inflight.py 16 64 256 512"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from run_config5 import load_l32, P  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16, 64, 256, 512]
    from plinopt_amd import CSEPlan, capi
    capi.check(capi.lib().plo_init(0))
    m, n, rp, c, v = load_l32()
    plan = CSEPlan(m, n, rp, c, v, P)
    os.environ["PLO_BIG_STATS"] = "1"
    for s in sorted(sizes, reverse=True):            # largest first: the workspace is allocated once
        os.environ["PLO_BIG_SLICES"] = str(s)
        a, mu = plan.cost_many(seed0=1, n=s)
        st = plan.last_stats
        print("in flight %4d: %8.1f ms  -> %6.1f candidates/s (grid %d); seed 1 -> %s" % (s, st["kernel_ms"], s / (st["kernel_ms"] * 1e-3), st["grid"], (a[0], mu[0])), flush=True)


if __name__ == "__main__":
    main()
