set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/t_r3_08.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_08.log | tail -5
[ $rc -ne 0 ] && exit $rc
timeout -k 10 240 python tools/sharded_overhead.py > gpurun_out/sharded_overhead_r3.json 2> gpurun_out/sharded_overhead_r3.err; echo "rc=$?"; grep workload gpurun_out/sharded_overhead_r3.json
