set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 240 python tools/sharded_overhead.py > gpurun_out/sharded_overhead_r3.json 2> gpurun_out/sharded_overhead_r3.err; echo "rc=$?"; cat gpurun_out/sharded_overhead_r3.json
timeout -k 10 240 python tools/sharded_overhead.py --zipf > gpurun_out/sharded_overhead_r3_zipf.json 2> gpurun_out/sharded_overhead_r3_zipf.err; echo "rc=$?"; cat gpurun_out/sharded_overhead_r3_zipf.json
BR_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy --no-legs > gpurun_out/bench_r3_forced_sharded.json 2> gpurun_out/bench_r3_forced_sharded.err; echo "rc=$?"; tail -3 gpurun_out/bench_r3_forced_sharded.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r3_forced_sharded.json')); print(d['ms_per_step'], d['launch_mode'])"
