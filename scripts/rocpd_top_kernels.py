"""Top kernels (average duration, calls) from a rocprofv3 `--kernel-trace --stats` run that wrote its default rocpd
database: python scripts/rocpd_top_kernels.py <path>_results.db"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
for name, calls, avg in con.execute("select name, total_calls, average from top_kernels limit 8"):
    print(f"{avg:9.3f} us x {calls:5d}  {name[:90]}")
