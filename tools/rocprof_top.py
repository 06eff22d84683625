#!/usr/bin/env python3
"""Prints the per-kernel summary of a rocprofv3 results database (the default output format of rocprofv3 in this image):
name, calls, total and average duration.  usage: tools/rocprof_top.py <dir or .db> [substring filter]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    print(f"# {db}")
    print(f"{'calls':>6} {'total ms':>10} {'avg ms':>10}  kernel")
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if flt in name:
            print(f"{calls:6d} {total / 1e3:10.3f} {avg / 1e3:10.3f}  {name[:110]}")
