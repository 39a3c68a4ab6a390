#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from a `*_pmc.csv` made by tests/rocpd_summary.py: HBM bytes per candidate of the config-5 kernel
with the hash of the kernel source they were measured on (bench.py marks the figure stale when the source has changed since).
usage: python tests/make_traffic_json.py profiles/r04a_32x32x32_pmc.csv 1024 "source text" [tag=r04] """
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(csv, cand, source, tag="r04"):
    vals = {}
    for ln in open(csv):
        t = ln.strip().split(",")
        if len(t) > 4 and t[2] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
            vals[t[2]] = float(t[4])
    old = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
    sha = hashlib.sha256(open(os.path.join(ROOT, "plinopt_amd", "csrc", "plo_cse_big.hip"), "rb").read()).hexdigest()[:16]
    old["_comment"] = ("HBM traffic per candidate from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB x 1024) of the search dispatch divided by its candidates; "
                       "kernel_source_sha16 = sha256 of plinopt_amd/csrc/plo_cse_big.hip at measurement time (bench.py reports the figure as stale when the source "
                       "differs). On gfx950 FETCH_SIZE reads half of the bytes of wide (16 B per lane) streaming reads (MI355X_MICROARCH.md, HBM): this kernel's reads are 4- and 8-byte "
                       "accesses except the image copies, so the counter is reported as it is and bench.py also gives the bound with the fetch side doubled. Earlier rounds: profiles/r01..r03_traffic.json.")
    old["32x32x32"] = {"fetch_bytes_per_candidate": vals["FETCH_SIZE"] * 1024.0 / cand, "write_bytes_per_candidate": vals["WRITE_SIZE"] * 1024.0 / cand,
                       "l2_requests_per_candidate": (vals.get("TCC_HIT_sum", 0.0) + vals.get("TCC_MISS_sum", 0.0)) / cand,
                       "l2_hit_rate": vals.get("TCC_HIT_sum", 0.0) / max(1.0, vals.get("TCC_HIT_sum", 0.0) + vals.get("TCC_MISS_sum", 0.0)),
                       "source": source, "kernel_source_sha16": sha}
    json.dump(old, open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w"), indent=1)
    print(json.dumps(old["32x32x32"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), sys.argv[3], *(sys.argv[4:5]))
