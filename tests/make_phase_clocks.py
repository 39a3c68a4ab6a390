#!/usr/bin/env python3
"""Phase clocks of plo::cse_big_kernel on config 5 (GPU box): a candidate alone on the chip (PLO_BIG_SLICES=1) and with
512 in flight (1024 candidates, 2 workgroups per CU), from the kernel's own wall-clock stamps (PLO_BIG_STATS=1: thread 0 of
the workgroup of the LAST finished candidate; us).  Writes profiles/<tag>_phase_clocks.json so that the figures quoted in
DESIGN.md can be recomputed.  usage: python tests/make_phase_clocks.py <tag> [ncand_loaded]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(ncand, env_extra):
    env = dict(os.environ, PLO_BIG_STATS="1", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_config5.py"), str(ncand)], capture_output=True, text=True, env=env)
    if r.returncode:
        raise SystemExit(r.stdout + r.stderr)
    out = {"ncand": ncand, "env": env_extra}
    m = re.search(r"candidates \d+ in [\d.]+ s \(kernel ([\d.]+) ms\): ([\d.]+) candidates/s; grid (\d+)", r.stdout)
    out.update(kernel_ms=float(m.group(1)), candidates_per_s=float(m.group(2)), grid=int(m.group(3)))
    m = re.search(r"steps (\d+), full scans (\d+), level rebuilds (\d+); phase us: level (\d+) select (\d+) rows (\d+) sweep1 (\d+) flush1 (\d+) sweep2 (\d+) flush2 (\d+) tail (\d+)", r.stderr)
    keys = ["steps", "merges", "level_rebuilds", "level_and_merges_us", "tie_pick_us", "row_search_us", "sweep_us", "flush_us", "sweep2_us", "flush2_us", "tail_us"]
    out["last_candidate"] = dict(zip(keys, map(int, m.groups())))
    m = re.search(r"per candidate ([\d.]+) merges \(([\d.]+) forced.*?, ([\d.]+) log records, ([\d.]+) hot-table updates; last candidate, merge us: hot->log (\d+), partition pass (\d+), sum \+ write back (\d+), window (\d+)", r.stderr)
    if m:
        out["merge"] = {"merges_per_candidate": float(m.group(1)), "forced": float(m.group(2)), "log_records": float(m.group(3)), "hot_updates": float(m.group(4)),
                        "hot_to_log_us": int(m.group(5)), "partition_pass_us": int(m.group(6)), "sum_write_back_us": int(m.group(7)), "window_us": int(m.group(8))}
    m = re.search(r"sum \+ write back of the last candidate: (\d+) groups; us: sum \(loads \+ table\) (\d+), scan \+ write back (\d+), clear \+ bounds (\d+)", r.stderr)
    if m:
        out["merge_sum"] = {"groups": int(m.group(1)), "sum_us": int(m.group(2)), "scan_write_back_us": int(m.group(3)), "clear_bounds_us": int(m.group(4))}
    m = re.search(r"image load \+ CSE phase (\d+) us, ProgramGen (\d+) us", r.stderr)
    if m:
        out["last_candidate"]["load_and_cse_us"] = int(m.group(1)); out["last_candidate"]["program_gen_us"] = int(m.group(2))
    out["stderr_tail"] = [ln for ln in r.stderr.splitlines() if ln.startswith("#")][-6:]
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    loaded = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    import hashlib
    src = open(os.path.join(ROOT, "plinopt_amd", "csrc", "plo_cse_big.hip"), "rb").read()
    res = {"kernel": "plo::cse_big_kernel<2,true>", "source_sha256_16": hashlib.sha256(src).hexdigest()[:16],
           "note": "wall-clock stamps of thread 0 (100 MHz counter) in the workgroup that finished last; level_and_merges_us holds the merges",
           "alone": run(1, {"PLO_BIG_SLICES": "1"}), "loaded": run(loaded, {})}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for d in ("gpurun_out", "profiles"):
        with open(os.path.join(ROOT, d, "%s_phase_clocks.json" % tag), "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
