out=$PWD/gpurun_out/r06zu; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/st -o s -- python scripts/ubench/bwd_bench.py c3 > $out/run.txt 2>&1
python - <<PY
import sqlite3,glob
for db in glob.glob("$out/st/**/*.db", recursive=True):
    con=sqlite3.connect(db)
    try:
        for r in con.execute("select name,total_calls,average from top_kernels limit 8"): print(r[0][:70], r[1], round(r[2]/1000,2),"us")
    except Exception as e: print(e)
PY
rm -rf $out/st
