#!/usr/bin/env python3
"""Generates tests/golden/l32cut_costs.json: (adds, muls) of the LITERAL CPU oracle (oracle/plo_oracle.c) for
seeds 1..16 on a row block of 32x32x32_15096_L mod 131071 that the literal oracle can still walk (it rescans the
whole pair map at every step: ~3 minutes per seed on rows [0,128), 7,824 non-zeros, 425,784 pair instances,
137,038 distinct triples, rows of 48 and 288 entries).  The HBM-resident kernel family (plo_cse_big.hip) and the
scalable host engine (plo_fast.hpp) are compared with these values; unlike tests/golden/config5_costs.json they do
not come from product code.  Run in the build container: python tests/golden/make_l32cut_costs.py  (~8 min, 8 threads)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plo_testlib import GOLDEN, OracleMatrix, l32_cut  # noqa: E402

P = 131071
WHICH = sys.argv[1] if len(sys.argv) > 1 else "A"
if WHICH == "A":          # rows [0,128), whole rows
    LO, HI, NSEEDS, TRUNC, OUT = 0, 128, 16, 0, "l32cut_costs.json"
else:                     # rows [0,24) and the first two rows of >= 768 entries, every row cut to its first 256 entries:
    LO, HI, NSEEDS, TRUNC, OUT = 0, 24, 8, 256, "l32cutB_costs.json"      # rows of 4 chunks of 64 lanes on the device
from plo_testlib import l32_rows, l32_cut_b  # noqa: E402
if WHICH == "A":
    m, n, rp, c, v = l32_cut(LO, HI, P)
else:
    m, n, rp, c, v = l32_cut_b(P)
M = OracleMatrix(m, n, rp, c, v, P)
t0 = time.time()
from concurrent.futures import ThreadPoolExecutor  # one seed per call: ctypes releases the GIL, 8 seeds run at once
with ThreadPoolExecutor(max_workers=8) as ex:
    res = list(ex.map(lambda s: M.cost_many(seed0=s, nseeds=1), range(1, 1 + NSEEDS)))
a, mu = [r[0][0] for r in res], [r[1][0] for r in res]
out = {"p": P, "matrix": "32x32x32_15096_L rows [%d,%d) (regenerated from the stored SLP)" % (LO, HI), "row_lo": LO, "row_hi": HI,
       "nnz": len(c), "seed0": 1, "adds": list(a), "muls": list(mu), "source": "oracle/plo_oracle.c (literal restatement)",
       "oracle_seconds": round(time.time() - t0, 1)}
out["truncate_rows_to"] = TRUNC
json.dump(out, open(os.path.join(GOLDEN, OUT), "w"), indent=0)
print(out)
