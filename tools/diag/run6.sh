set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_exchange.py tests/test_gpu_sharded.py -x -q -m gpu -k "exchange or dedup or five or surface" > gpurun_out/t_r3_06.log 2>&1; echo "tests rc=$?" ; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_06.log | tail -30
