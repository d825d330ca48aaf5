#!/usr/bin/env python
"""Per-kernel averages of the counters of one rocprofv3 --pmc run:  pmc_dump.py <dir> [kernel substring]"""
import glob
import sqlite3
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection "
                           "group by kernel_name,counter_name").fetchall()
    except Exception as exc:  # noqa: BLE001
        print(db, exc)
        continue
    for name, ctr, val, n in rows:
        if pat in name:
            print(f"{name.split('(')[0][-60:]:60s} {ctr:32s} {val:16.1f} x{n}")
