#!/bin/bash
# GPU box: calibrate FETCH_SIZE / WRITE_SIZE against known byte counts per access pattern (tools/ubench/fetch_calib.hip)
#   -> gpurun_out/r06_fetch_calibration.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/ubench/fetch_calib.hip || exit 1
W=/tmp/fcal; rm -rf $W; mkdir -p $W
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C -d $W/$C -o p -- /tmp/fetch_calib 1024 > $W/$C.log 2>&1; done
python - "$W" > $O/r06_fetch_calibration.txt <<'PY'
import re, sqlite3, sys
w = sys.argv[1]
known = dict(zip(*[iter(open(w + "/FETCH_SIZE.log").read().split("known_kib", 1)[1].split())] * 2))
print("# FETCH_SIZE / WRITE_SIZE (KiB per dispatch, rocprofv3 --pmc, average of 3 dispatches; the first one reads a cold buffer) against the KNOWN KiB each kernel moves once")
print("# tools/ubench/fetch_calib.hip, 1 GiB buffer (beyond L2 + the 256 MB last-level cache)")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    c = sqlite3.connect("%s/%s/p_results.db" % (w, counter))
    for kn, n, v in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        short = re.sub(r"\(.*", "", kn).split()[-1]
        k = float(known.get(short, 0))
        relevant = short.startswith("read") == (counter == "FETCH_SIZE")
        print("%-11s %-10s dispatches %d  counter %12.0f KiB   known %10.0f KiB   counter/known %.3f%s" % (
            counter, short, n, v, k, v / k if k else 0, "" if relevant else "   (the other direction)"))
PY
cat $O/r06_fetch_calibration.txt
