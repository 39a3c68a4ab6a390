#!/usr/bin/env python3
"""profiles/r04_issue.json entry (the file starts as a copy of r03_issue.json: entries of unchanged kernels stay) from a rocprofv3 --pmc summary (tests/rocpd_summary.py *_pmc.csv):
   python tests/make_issue_json.py <workload> <kernel substring> <profiles/xxx_pmc.csv> "<source note>"
Rates are per clock and compute unit (256 CUs); cycles = SQ_BUSY_CYCLES / 32 (MI355X_MICROARCH.md, profiling section).
Every entry carries the hash of the kernel sources it was measured on: bench.py marks the figure stale when they have changed."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = {"tril": ["plo_tril.hip"], "cob": ["plo_cob.hip"], "kmethod": ["plo_kmethod.hip", "plo_cse_wave.hip"]}      # every other workload: the wave kernel


def sources_sha16(name):
    h = hashlib.sha256()
    for f in SOURCES.get(name, ["plo_cse_wave.hip"]):
        h.update(open(os.path.join(ROOT, "plinopt_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main(name, kern, csv, note):
    avg = {}
    for ln in open(csv):
        f = ln.strip().split(",")
        if len(f) >= 5 and kern in f[1]:
            avg[f[2]] = float(f[4])
    cyc = avg["SQ_BUSY_CYCLES"] / 32.0
    e = {"cycles_per_launch": cyc,
         "valu_per_clk_cu": avg["SQ_INSTS_VALU"] / cyc / 256, "salu_per_clk_cu": avg["SQ_INSTS_SALU"] / cyc / 256,
         "lds_per_clk_cu": avg["SQ_INSTS_LDS"] / cyc / 256, "vmem_per_clk_cu": (avg.get("SQ_INSTS_VMEM_RD", 0) + avg.get("SQ_INSTS_VMEM_WR", 0)) / cyc / 256,
         "source": note, "kernel_source_sha16": sources_sha16(name)}
    e["bound_unit"] = "scalar" if e["salu_per_clk_cu"] / 1.0 >= e["valu_per_clk_cu"] / 2.0 else "vector"
    p = os.path.join(ROOT, "profiles", "r04_issue.json")
    prev = os.path.join(ROOT, "profiles", "r03_issue.json")
    d = json.load(open(p)) if os.path.exists(p) else json.load(open(prev)) if os.path.exists(prev) else {"_comment": "Issue-rate view of the kernels that are not HBM-bound (rocprofv3 --pmc SQ_INSTS_* / SQ_BUSY_CYCLES, per launch; cycles = SQ_BUSY_CYCLES / 32 shader engines; rates per clock and compute unit, 256 CUs). Peaks used by bench.py: 1 scalar and 2 vector wave-instructions per clock and CU. kernel_source_sha16: sha256 of the kernel source(s) at measurement time."}
    d[name] = e
    json.dump(d, open(p, "w"), indent=1)
    print(name, json.dumps(e))


if __name__ == "__main__":
    main(*sys.argv[1:5])
