set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sparse_optim.py tests/test_gpu_neumf.py tests/test_gpu_twotower_bpr.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/t_r3_03.log 2>&1; echo "tests rc=$?" ; tail -12 gpurun_out/t_r3_03.log
