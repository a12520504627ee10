set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_rows_f.py tests/test_gpu_twotower_bpr.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/t_r3_09.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_09.log | tail -25
