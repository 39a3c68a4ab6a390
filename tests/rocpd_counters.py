#!/usr/bin/env python3
"""Print the counters of every rocprofv3 --pmc pass found below a directory (rocpd SQLite).
usage: python tests/rocpd_counters.py gpurun_out/prof_r01e_latency"""
import glob
import sqlite3
import sys

for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error as e:
        print(db, "unreadable:", e); continue
    for k, c, n, s in rows:
        if "plo::" in k:
            print("%s,%s,%s,%d,%.6g" % (db.split("/")[-3] if db.count("/") > 2 else db, k.split("(")[0], c, n, s))
