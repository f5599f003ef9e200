#!/bin/bash
# GPU box: refresh the counter-derived HBM traffic of the headline kernel (bench.py's `roofline.traffic`).
#   tools/r05_traffic.sh [tag]      -> gpurun_out/<tag>_wavenet_traffic.json (copy it to profiles/ to have bench.py use it)
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_MFMA), as MI355X_MICROARCH.md prescribes; the JSON records the
# kernel name and the MFMA instruction count per launch, and bench.py REFUSES a file whose kernel name or
# SQ_INSTS_MFMA x 4096 disagrees (> 1 %) with the executed flops it computes for the kernel it is about to time.
set -u
TAG=${1:-r05}; PREC=${2:-f32}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=/tmp/traffic_$TAG; rm -rf $W; mkdir -p $W
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-full-loop --precision $PREC"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_MFMA; do
  rocprofv3 --pmc $C -d $W/$C -o p -- $CMD > $W/$C.log 2>&1
done
python - "$W" "$OUT/${TAG}_wavenet_traffic_${PREC}.json" "$PREC" <<'PY'
import json, sqlite3, sys
w, dst, prec = sys.argv[1], sys.argv[2], sys.argv[3]
# flops of one MFMA instruction: v_mfma_f32_32x32x2_f32 = 4096; v_mfma_f32_32x32x16_bf16 = 32768, six of them per
# fp32-equivalent product term (bf16x6) or three (bf16x3)
per_inst = {"f32": 4096.0, "bf16x6": 32768.0 / 6.0, "bf16x3": 32768.0 / 3.0, "f16x3": 32768.0 / 3.0}[prec]
def avg(counter):
    c = sqlite3.connect("%s/%s/p_results.db" % (w, counter))
    rows = list(c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and "
                          "kernel_name like '%wn_layer_%' group by kernel_name order by count(*) desc", (counter,)))
    return rows[0]
(kn, n, fetch), (_, _, write), (_, _, mfma) = avg("FETCH_SIZE"), avg("WRITE_SIZE"), avg("SQ_INSTS_MFMA")
short = kn.split("dws::", 1)[1].split("(")[0].replace(" ", "")      # void dws::NAME<...>(dws::Args, ...) -> NAME<...>
json.dump({"kernel": short, "dispatches": n,
           "source": "tools/r05_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_MFMA, separate passes, bench.py --steps 3",
           "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
           "gfx950_correction": "FETCH_SIZE reports 1/2 of a wide coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM "
                                "section): read bytes = 2 * FETCH_SIZE * 1024",
           "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
           "precision": prec, "sq_insts_mfma_per_launch": mfma, "fp32_equivalent_flops_per_mfma_inst": per_inst,
           "executed_flops_per_launch_from_counter": mfma * per_inst}, open(dst, "w"), indent=2)
print(open(dst).read())
PY
rm -rf $W
