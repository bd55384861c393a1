import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name,total_calls,average from top_kernels limit 8"):
    print(f"{r[2]:9.3f} us x {r[1]:5d}  {r[0][:90]}")
