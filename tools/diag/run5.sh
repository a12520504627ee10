set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_nccl_world1.py -x -q -m gpu > gpurun_out/t_r3_05.log 2>&1; echo "tests rc=$?" ; tail -25 gpurun_out/t_r3_05.log
