#!/bin/bash
out=gpurun_out/r05zs
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python scripts/ubench/lmi_sweep.py > $out/sweep_under_profiler.txt 2> $out/stats.err
python - <<'PY'
import sqlite3, json
con = sqlite3.connect("gpurun_out/r05zs/stats/s_results.db")
rows = []
for r in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
    rows.append({"name": r[0][:110], "calls": r[1], "total_us": r[2], "avg_us": r[3], "pct": r[4]})
json.dump({"command": "rocprofv3 --kernel-trace --stats -- python scripts/ubench/lmi_sweep.py   (final library: S(v) by the library GEMM from 32 generators on)", "top_kernels": rows}, open("gpurun_out/r05zs/lmi_sweep_products_rocprofv3.json", "w"), indent=1)
for r in rows: print(r["name"][:90], r["calls"], round(r["avg_us"],1), round(r["pct"],1))
PY
rm -rf $out/stats
