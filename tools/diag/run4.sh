set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mode in exact fast; do
  python bench.py --dim 128 --users 12500000 --items 1250000 --int64 --steady-state-lags --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-lazy --replay $mode > gpurun_out/bench_r3_c5_$mode.json 2> gpurun_out/bench_r3_c5_$mode.err || { tail -5 gpurun_out/bench_r3_c5_$mode.err; exit 1; }
done
python - <<'PY'
import json
for f in ("c5_exact","c5_fast"):
    d=json.load(open(f"gpurun_out/bench_r3_{f}.json"))
    print(f, "ms/step", round(d["ms_per_step"],4), "graph", d["whole_step_graph"] and round(d["whole_step_graph"]["ms_per_step"],4), "roofline", d["roofline"]["kernel"][:30], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"],1), d["launch_mode"][:60])
    for k,v in d["kernels"].items(): print("   ", k[:70], round(v["us"],1), round(v.get("frac",0) or 0,3))
PY
