set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/t_r3_04.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/t_r3_04.log
[ $rc -ne 0 ] && exit $rc
for mode in exact fast; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy --no-legs --replay $mode > gpurun_out/bench_r3_c2_$mode.json 2> gpurun_out/bench_r3_c2_$mode.err || exit 1
  python bench.py --dim 128 --users 12500000 --items 1250000 --int64 --steady-state-lags --steps 20 --warmup 5 --no-cpu-baseline --no-legs --no-lazy --replay $mode > gpurun_out/bench_r3_c5_$mode.json 2> gpurun_out/bench_r3_c5_$mode.err || exit 1
done
python - <<'PY'
import json
for f in ("c2_exact","c2_fast","c5_exact","c5_fast"):
    d=json.load(open(f"gpurun_out/bench_r3_{f}.json"))
    print(f, "ms/step", round(d["ms_per_step"],4), "graph", d["whole_step_graph"] and round(d["whole_step_graph"]["ms_per_step"],4), "roofline", d["roofline"]["kernel"][:30], round(d["roofline"]["frac"],3))
    for k,v in d["kernels"].items(): print("   ", k[:70], round(v["us"],1), round(v.get("frac",0) or 0,3))
    if d.get("steady_state_lags"): print("   seeded", {k:(round(v["mean_lag"],1), round(v["frac_capped_by_flush"],3)) for k,v in d["steady_state_lags"].items() if isinstance(v,dict)})
PY
