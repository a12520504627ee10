set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_neumf.py tests/test_gpu_sparse_optim.py tests/test_gpu_twotower_bpr.py -x -q -m gpu > gpurun_out/t_r3_01.log 2>&1; echo "tests rc=$?" ; tail -5 gpurun_out/t_r3_01.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "12500000 or neumf_steps" > gpurun_out/t_r3_02.log 2>&1; echo "fullsize rc=$?"; tail -5 gpurun_out/t_r3_02.log
