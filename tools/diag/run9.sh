set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_sh
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sh -o t -- python3 tools/sharded_overhead.py --local-only > gpurun_out/prof_sh.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/prof_sh -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f | cut -c1-150
