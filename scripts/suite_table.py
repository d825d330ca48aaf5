import sqlite3,sys
con=sqlite3.connect(sys.argv[1])
for r in con.execute("select name,total_calls,average from top_kernels"):
    print(f"{r[0][:90]:90s} {r[1]:5d} {r[2]:9.2f}")
