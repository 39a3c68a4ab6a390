#!/usr/bin/env python3
"""Generates tests/golden/longrow_costs.json: (adds, muls) of the LITERAL CPU oracle (oracle/plo_oracle.c) on a synthetic matrix whose
rows exceed 512 entries (plo_testlib.longrow_matrix: 9 rows of 461-627 entries over 640 columns, shared support, four residues).
The cuts of 32x32x32_15096_L that the literal oracle can walk stop at rows of 288 / 256 entries (l32cut*_costs.json); longer rows of
the metric's input are covered only through the build's own scalable engine (plo_fast.hpp).  This fixture lets rows of more than
512 entries -- several 64-lane chunks per row, and with PLO_BIG_FWIN several windows of the flat sweep -- meet the literal oracle.
The oracle re-counts every pair of a rewritten row at every step (rows x L^2 map operations): tens of minutes per seed; run in the
build container: python tests/golden/make_longrow_costs.py [nseeds=4]   (one thread per seed)."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from plo_testlib import GOLDEN, OracleMatrix, longrow_matrix  # noqa: E402

P = 131071
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m, n, rp, c, v = longrow_matrix(P)
M = OracleMatrix(m, n, rp, c, v, P)
t0 = time.time()
with ThreadPoolExecutor(max_workers=NSEEDS) as ex:
    res = list(ex.map(lambda s: M.cost_many(seed0=s, nseeds=1), range(1, 1 + NSEEDS)))
out = {"p": P, "matrix": "plo_testlib.longrow_matrix(131071): %d rows of %d..%d entries, %d columns" % (m, min(rp[i + 1] - rp[i] for i in range(m)), max(rp[i + 1] - rp[i] for i in range(m)), n),
       "nnz": len(c), "row_lengths": [rp[i + 1] - rp[i] for i in range(m)], "seed0": 1, "adds": [r[0][0] for r in res], "muls": [r[1][0] for r in res],
       "source": "oracle/plo_oracle.c (literal restatement)", "oracle_seconds": round(time.time() - t0, 1)}
json.dump(out, open(os.path.join(GOLDEN, "longrow_costs.json"), "w"), indent=0)
print(out)
