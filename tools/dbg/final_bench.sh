SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "rc=$? wall=$SECONDS s"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/final_bench.json") if l.startswith("{")][0])
print(d["metric"][:60], d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["frac"], d["roofline"]["traffic"])
print({k: round(v["ms_per_step"],3) for k,v in d.items() if k.startswith("extra_") and isinstance(v,dict) and "ms_per_step" in v})
for k,v in d["extra_configs"].items(): print(k, round(v.get("ms_per_step",0),3), {kk: round(vv["ms_per_step"],3) for kk,vv in v.items() if kk.startswith("extra_")}, v.get("error"))
PY
