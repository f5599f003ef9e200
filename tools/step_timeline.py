#!/usr/bin/env python
"""Launch-by-launch timeline of the LAST step of a traced run (rocprofv3 --kernel-trace --output-format csv):

  python tools/step_timeline.py <kernel_trace.csv> <steps in the trace> [--seq]

prints: the step's span, busy time, idle time (gaps between consecutive launches on the device), the gaps grouped by the
kernel that FOLLOWS them, the small launches (< 12 us) grouped by kernel, and with --seq the whole sequence.
A step boundary is the optimizer's first multi_tensor_apply launch after a run of non-optimizer launches."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("dws::", "")
    return n[:90]


def main():
    path, nsteps = sys.argv[1], int(sys.argv[2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # step boundaries: the first Adam launch after something that is not Adam
    is_opt = [("multi_tensor_apply" in n) for _, _, n in rows]
    bounds = [i for i in range(1, len(rows)) if is_opt[i] and not is_opt[i - 1]]
    # the end of a step = the last consecutive optimizer launch after bounds[k]
    ends = []
    for b in bounds:
        j = b
        while j + 1 < len(rows) and is_opt[j + 1]:
            j += 1
        ends.append(j)
    ends = ends[-(nsteps + 1):] if len(ends) > nsteps else ends
    if len(ends) < 2:
        print("fewer than two optimizer phases in the trace")
        return
    lo, hi = ends[-2] + 1, ends[-1]
    step = rows[lo:hi + 1]
    span = (step[-1][1] - step[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in step) / 1e3
    print("launches %d  span %.1f us  sum of kernel time %.1f us" % (len(step), span, busy))
    gaps = defaultdict(lambda: [0, 0.0])
    idle = 0.0
    cur_end = step[0][1]
    for s, e, n in step[1:]:
        g = (s - cur_end) / 1e3
        if g > 0:
            idle += g
            gaps[short(n)][0] += 1
            gaps[short(n)][1] += g
        cur_end = max(cur_end, e)
    print("idle between launches %.1f us (%.1f %% of the span)" % (idle, 100 * idle / span))
    print("\n-- idle time by the kernel that follows the gap (top 25)")
    for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%8.1f us %5d gaps  %s" % (t, c, n))
    small = defaultdict(lambda: [0, 0.0])
    for s, e, n in step:
        if e - s < 12000:
            small[short(n)][0] += 1
            small[short(n)][1] += (e - s) / 1e3
    print("\n-- launches shorter than 12 us: %d, %.1f us in total" % (sum(c for c, _ in small.values()), sum(t for _, t in small.values())))
    for n, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%8.1f us %5d  %s" % (t, c, n))
    if "--seq" in sys.argv:
        print("\n-- sequence (gap before | duration | kernel)")
        cur_end = step[0][0]
        for s, e, n in step:
            print("%7.1f %8.1f  %s" % ((s - cur_end) / 1e3, (e - s) / 1e3, short(n)))
            cur_end = max(cur_end, e)


if __name__ == "__main__":
    main()
