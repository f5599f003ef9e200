#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs into text for profiles/.

  python tools/rocpd_summary.py stats  gpurun_out/prof_stats/stats_results.db
  python tools/rocpd_summary.py pmc    gpurun_out/prof_pmc1/pmc1_results.db [kernel-substring]
"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    print(f"{'kernel':<100} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}")
    for name, calls, tot, avg, pct in c.execute("select * from top_kernels"):
        print(f"{name[:100]:<100} {calls:>7} {tot:>12.1f} {avg:>10.2f} {pct:>6.2f}")


def pmc(db, sub=""):
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
         "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name")
    print(f"{'kernel':<60} {'counter':<32} {'dispatches':>10} {'sum':>18} {'avg/dispatch':>18} {'avg_dur_ns':>12}")
    for k, n, cnt, s, a, d in c.execute(q, (f"%{sub}%",)):
        print(f"{k[:60]:<60} {n:<32} {cnt:>10} {s:>18.1f} {a:>18.1f} {d or 0:>12.0f}")


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    if mode == "stats":
        stats(db)
    else:
        pmc(db, sys.argv[3] if len(sys.argv) > 3 else "")
