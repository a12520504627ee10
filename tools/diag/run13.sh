set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bpr5
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bpr5 -o t --output-format csv -- python3 tools/diag/bpr_stats.py > gpurun_out/prof_bpr5.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/prof_bpr5 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-160; else echo "no stats file"; fi
rm -f gpurun_out/prof_bpr5/*kernel_trace.csv
