#!/usr/bin/env python3
"""Generates tests/golden/config5_costs.json: (adds, muls) of seeds 1..40 on 32x32x32_15096_L mod 131071
from the build's scalable host engine (bin/optimizer --replay --engine fast).  Each program is verified
with bin/SLPchecker against the regenerated matrix before its cost is recorded.  ~25 s per seed and core
(6 worker processes: ~4 minutes)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DATA = os.path.join(ROOT, "tests", "golden", "data")
P = 131071
sms = "/tmp/plo_l32_%d.sms" % os.getpid()
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "plinopt_amd", "csrc", "host")])
open(sms, "w").write(subprocess.run([os.path.join(ROOT, "bin", "SLPchecker"), "-q", str(P), os.path.join(DATA, "32x32x32_15096_L.slp")],
                                    capture_output=True, text=True, check=True).stdout)
out = {"p": P, "matrix": "32x32x32_15096_L (regenerated from the stored SLP)", "costs": {}}


def one(seed):
    r = subprocess.run([os.path.join(ROOT, "bin", "optimizer"), "-q", str(P), "--replay", "--seed", str(seed), sms], capture_output=True, text=True, check=True)
    a = int(re.search(r"# (\d+)\tadditions", r.stderr).group(1)); mu = int(re.search(r"# (\d+)\tmultiplications", r.stderr).group(1))
    chk = subprocess.run([os.path.join(ROOT, "bin", "SLPchecker"), "-q", str(P), "-M", sms], input=r.stdout, capture_output=True, text=True)
    assert chk.returncode == 0 and "SUCCESS" in chk.stderr and ("%d,%d" % (a, mu)) in chk.stderr, chk.stderr
    print(seed, a, mu, flush=True)
    return seed, a, mu


from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(max_workers=6) as ex:
    for seed, a, mu in ex.map(one, range(1, 41)):
        out["costs"][str(seed)] = [a, mu]
os.unlink(sms)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config5_costs.json"), "w"), indent=0)
