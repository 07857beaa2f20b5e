#!/usr/bin/env python3
"""Print / save the top kernels of a rocprofv3 --kernel-trace --stats sqlite output."""
import glob, json, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = [dict(zip(("name", "calls", "total_us", "avg_us", "pct"), r)) for r in con.execute("select * from top_kernels limit 14")]
for r in rows:
    r["name"] = r["name"][:90]
json.dump(rows, open(sys.argv[2], "w"), indent=1)
for r in rows:
    print("%-90s %6d %10.1f %9.2f %5.1f" % (r["name"], r["calls"], r["total_us"], r["avg_us"], r["pct"]))
