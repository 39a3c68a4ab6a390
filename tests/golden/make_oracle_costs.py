#!/usr/bin/env python3
"""Generates tests/golden/oracle_costs.json: (adds, muls) of the CPU oracle for a fixed seed
range on a few reference matrices.  These vectors come from the build's own oracle (the genuine
reference cannot be built here), so they pin the oracle against drift, not against the reference."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plo_testlib import DATA, GOLDEN, OracleMatrix  # noqa: E402

P = 131071
names = ["2x2x2_7_Winograd_L.sms", "cyclic.sms", "4x4x4_49_156_L.sms", "4x4x4_49_156_P.sms",
         "2x2x2_7_DPS-accurate_L.sms", "4x4x4_48_rational_P.sms", "3o3o6_Toom4_P.sms", "3x3x6_40_R.sms"]
out = {"p": P, "rng": "state0=1+splitmix64(seed)%(2^31-2); next=950706376*s%(2^31-1); pick=next%ties", "matrices": {}}
for nme in names:
    M = OracleMatrix.from_sms(os.path.join(DATA, nme), P)
    a, mu = M.cost_many(seed0=0, nseeds=64)
    out["matrices"][nme] = {"seed0": 0, "adds": a, "muls": mu}
json.dump(out, open(os.path.join(GOLDEN, "oracle_costs.json"), "w"), indent=0)
print("written")
