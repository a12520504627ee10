set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_r3_11.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_11.log | tail -6
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3_a.json 2> gpurun_out/bench_r3_a.err; echo "bench rc=$?"; tail -4 gpurun_out/bench_r3_a.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r3_a.json"))
print("value", d["value"], "ms", d["ms_per_step"], "graph", d["whole_step_graph"], "roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"])
for k,v in d["kernels"].items(): print("   ", k[:70], round(v["us"],1), round(v.get("frac",0) or 0,3))
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("ms_per_step","value","us","frac_read")}) for k,v in d["legs"].items()})
print(d["cpu_baseline"])
PY
