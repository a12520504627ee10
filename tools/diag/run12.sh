set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_twotower_bpr.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bpr or twotower" > gpurun_out/t_r3_10.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_10.log | tail -25
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python - > gpurun_out/legs_r3.json 2> gpurun_out/legs_r3.err <<'PY'
import sys, json, importlib, torch
sys.argv = ["bench.py"]
import bench
dev = torch.device("cuda:0")
ops = importlib.import_module("binary-recommendation_amd.ops")
out = {"bpr": bench.bpr_leg(dev, 1_000_000, 100_000, 64, 65536, 7), "twotower": bench.twotower_leg(ops, dev, 1_000_000, 100_000, 64, 64, 8192, 11)}
print(json.dumps(out))
PY
echo "legs rc=$?"; tail -3 gpurun_out/legs_r3.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/legs_r3.json"))
b=d["bpr"]; print("bpr", {k:(round(v["ms_per_step"],4), v.get("hipgraph_replay")) for k,v in b.items() if isinstance(v,dict)})
t=d["twotower"]; print("twotower", t["ms_per_step"], t["hipgraph_replay"], t["config4_per_rank_stripe"], t["inbatch_softmax"]["frac"])
PY
