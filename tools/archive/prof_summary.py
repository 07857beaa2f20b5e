#!/usr/bin/env python3
"""Summarise the rocprofv3 sqlite outputs of tools/prof_counters.sh into one JSON (committed under profiles/)."""
import glob, json, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
out = {}
con = sqlite3.connect(glob.glob(src + "/kt/*.db")[0])
out["kernel_trace_stats"] = [dict(zip(("name", "calls", "total_us", "avg_us", "pct"), r))
                             for r in con.execute("select * from top_kernels limit 3")]
for db in sorted(glob.glob(src + "/pmc*/*.db")):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "group by kernel_name, counter_name"):
        if "lqr_step" in r[0]:
            out.setdefault("pmc_avg_per_dispatch", {}).setdefault(r[0].split("(")[0][-40:], {})[r[1]] = r[3]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
