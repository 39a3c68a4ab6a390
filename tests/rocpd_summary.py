#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd SQLite) outputs of tests/profile_r01.sh into text files under profiles/.
usage: python tests/rocpd_summary.py gpurun_out/prof_r01_winograd profiles/r01_winograd"""
import glob
import json
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(sql).fetchall()
    return [d[0] for d in cur.description], rows


def main(src, dst):
    out = {}
    dbs = glob.glob(src + "/trace/*.db")
    lines = ["# rocprofv3 --kernel-trace --stats (rocpd top_kernels view); durations in us",
             "name,total_calls,total_duration_us,average_us,percentage"]
    for db in dbs:
        cols, rows = q(db, "select name,total_calls,total_duration,average,percentage from top_kernels")
        for r in rows:
            lines.append("%s,%d,%.3f,%.3f,%.3f" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
        cols, rows = q(db, "select name, count(*), min(end-start), max(end-start), avg(end-start) from kernels group by name")
        lines.append("# per-dispatch durations (ns): name,calls,min,max,avg")
        for r in rows:
            lines.append("%s,%d,%d,%d,%.1f" % (r[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
        cols, rows = q(db, "select name, grid_x, workgroup_x, lds_size, count(*), avg(end-start) from kernels group by name, grid_x order by 6 desc")
        lines.append("# by grid: name,grid_x,wg_x,lds,calls,avg_ns")
        for r in rows:
            lines.append(",".join(str(x).replace(",", ";") for x in r))
    open(dst + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")
    pl = ["# rocprofv3 --pmc passes (separate runs); per kernel: pass,kernel,counter,dispatches,avg_per_dispatch,sum,max",
          "# FETCH_SIZE/WRITE_SIZE are in KiB as reported; gfx950: FETCH_SIZE reads 1/2 of wide streaming reads (MI355X_MICROARCH.md, HBM)"]
    for sub in ("pmc_sq", "pmc_tcc", "pmc_fetch", "pmc_write"):
        for db in glob.glob(src + "/%s/*.db" % sub):
            cols, rows = q(db, "select kernel_name, counter_name, count(*), avg(value), sum(value), max(value) from counters_collection "
                               "group by kernel_name, counter_name")
            for r in rows:
                if "plo::" in r[0]:
                    pl.append("%s,%s,%s,%d,%.3f,%.3f,max=%.3f" % (sub, r[0].replace(",", ";"), r[1], r[2], r[3], r[4], r[5]))
    open(dst + "_pmc.csv", "w").write("\n".join(pl) + "\n")
    for f in glob.glob(src + "/bench_*.json"):
        try:
            out[f.split("/")[-1]] = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            pass
    json.dump(out, open(dst + "_bench_under_profiler.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
