#!/bin/bash
# GPU box: counter-derived HBM traffic of the SaShiMi step's two kernel families (S4 tails, fused FFT convolution) for
# bench.py's `roofline.traffic` of the C3 / C4 legs.   tools/r06_traffic_sashimi.sh <config> [tag] [precision]
#   -> gpurun_out/<tag>_sashimi_traffic_<config>_<precision>.json  (copy to profiles/ to have bench.py use it)
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_MFMA) as MI355X_MICROARCH.md prescribes.  The JSON keeps, per
# family, the dispatch-weighted average per launch and the MFMA instruction count; bench.py refuses the file when the
# config differs or SQ_INSTS_MFMA x 4096 of the tail family is more than 5 % off the tail flops it computes (a file
# measured on other kernels -- e.g. the bf16 split tails -- does not pass).
set -u
CFG=$1; TAG=${2:-r06}; PREC=${3:-bf16x6}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=/tmp/traffic_ss_$TAG; rm -rf $W; mkdir -p $W
CMD="python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-full-loop --precision $PREC"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_MFMA; do
  rocprofv3 --pmc $C -d $W/$C -o p -- $CMD > $W/$C.log 2>&1
done
python - "$W" "$OUT/${TAG}_sashimi_traffic_${CFG}_${PREC}.json" "$CFG" "$PREC" <<'PY'
import json, sqlite3, sys
w, dst, cfg, prec = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
def fam(counter, like):
    c = sqlite3.connect("%s/%s/p_results.db" % (w, counter))
    rows = list(c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? and "
                          "kernel_name like ? group by kernel_name", (counter, like)))
    n = sum(r[1] for r in rows)
    return n, (sum(r[2] for r in rows) / n if n else 0.0), {r[0].split("dws::", 1)[-1].split("(")[0].replace(" ", ""): r[1] for r in rows}
out = {"config": cfg, "precision": prec, "source": "tools/r06_traffic_sashimi.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_MFMA, separate passes, "
                                "bench.py --steps 3 --warmup 1",
       "gfx950_correction": "FETCH_SIZE reports 1/2 of a wide coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM section): "
                            "read bytes = 2 * FETCH_SIZE * 1024", "families": {}}
for name, like in (("s4_tail", "%s4_tail%"), ("fftconv", "%fftconv_%")):
    n, fetch, kern = fam("FETCH_SIZE", like)
    _, write, _ = fam("WRITE_SIZE", like)
    _, mfma, _ = fam("SQ_INSTS_MFMA", like)
    out["families"][name] = {"dispatches": n, "kernels": kern, "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
                             "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "sq_insts_mfma_per_launch": mfma}
json.dump(out, open(dst, "w"), indent=2)
print(open(dst).read())
PY
rm -rf $W
