set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nccl_world1.py -x -q -m gpu -s > gpurun_out/t_r3_07.log 2>&1; echo "tests rc=$?" ; grep -v "Gloo\|amdgpu.ids\|socket.cpp" gpurun_out/t_r3_07.log | tail -30
