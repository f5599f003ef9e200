#!/usr/bin/env python3
"""usage: tools/hip_api_summary.py <rocprofv3 --hip-trace results.db>: HIP API call counts and the hipMemcpyAsync calls by size."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(regions)")]
print("regions columns:", cols)
rows = c.execute("select name, count(*), sum(end - start) from regions group by name order by 2 desc limit 25").fetchall()
for n, k, t in rows:
    print(f"{n:50s} {k:8d} {t / 1e6:10.2f} ms")
acols = [r[1] for r in c.execute("pragma table_info(region_args)")]
print("region_args columns:", acols)
try:
    q = ("select a.value, count(*) from regions r join region_args a on a.id = r.id "
         "where r.name = 'hipMemcpyAsync' and a.name in ('sizeBytes', 'size') group by a.value order by 2 desc limit 30")
    for v, k in c.execute(q):
        print("hipMemcpyAsync size", v, "x", k)
except Exception as e:
    print("args query failed:", e)
