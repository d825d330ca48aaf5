"""rocprofv3 --kernel-trace --stats of scripts/ubench/bwd_profile.py -> a small JSON (run on the GPU box):
    python scripts/ubench/bwd_profile_summary.py c3 out.json"""
import glob, json, os, sqlite3, subprocess, sys, tempfile
cfg, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = tempfile.mkdtemp(prefix="bwdprof_")
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "bwd", "--", sys.executable,
                os.path.join(root, "scripts", "ubench", "bwd_profile.py"), cfg], env=env, check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
rows = [{"name": r[0][:110], "calls": r[1], "avg_us": r[3], "pct": r[4]}
        for r in sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6")]
json.dump({"command": f"rocprofv3 --kernel-trace --stats -- python scripts/ubench/bwd_profile.py {cfg}",
           "note": "200 backward calls at B = 262144 (fp32) after one tracked forward; the kernels of one backward call",
           "kernel_stats": rows}, open(out, "w"), indent=1)
print(json.dumps(rows[:4]))
