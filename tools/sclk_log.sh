#!/bin/bash
# usage: tools/sclk_log.sh <outfile> -- <command...>   samples the GPU clocks (rocm-smi -g / power) every 0.25 s while
# the command runs and writes min / mean / max sclk of the samples taken at >= 50 % GPU use.
OUT=$1; shift; shift
( while true; do rocm-smi -g -u -P --csv 2>/dev/null | tail -n +2 | head -1; sleep 0.25; done ) > /tmp/sclk_raw.csv &
LOGPID=$!
"$@"
RC=$?
kill $LOGPID 2>/dev/null
python - "$OUT" <<'PY'
import re, sys
rows = [l.strip() for l in open('/tmp/sclk_raw.csv') if l.strip()]
clk = []
for l in rows:
    m = re.findall(r'\((\d+)Mhz\)', l)
    nums = re.findall(r'(?<![\w.])(\d+(?:\.\d+)?)(?![\w.])', l)
    if m:
        clk.append((int(m[0]), l))
busy = [c for c, l in clk]
with open(sys.argv[1], 'w') as f:
    f.write("samples %d\n" % len(clk))
    if busy:
        top = sorted(busy)
        f.write("sclk MHz: min %d median %d max %d\n" % (top[0], top[len(top) // 2], top[-1]))
    f.write("raw (every 8th):\n" + "\n".join(rows[::8]) + "\n")
PY
exit $RC
