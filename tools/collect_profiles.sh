#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + two separate PMC passes (FETCH_SIZE / WRITE_SIZE
# cannot share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots") of the default bench workload.
# Outputs land in $GRAFT_REPO_ROOT/gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/.
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-lazy --no-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err && echo "trace ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err && echo "fetch ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err && echo "write ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $OUT/sq -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err && echo "sq ok"
# one replayed step's timeline (start / duration / gap per kernel) before the traces go
python3 $GRAFT_REPO_ROOT/tools/timeline.py $OUT/trace -30 > $OUT/graph_replay_timeline_$TAG.txt 2>&1 && echo "timeline ok"
# keep only what is needed (the raw per-dispatch counter CSVs are large)
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $OUT $TAG > $OUT/summary_$TAG.json && echo "summary ok"
rm -f $OUT/fetch/*counter_collection.csv $OUT/write/*counter_collection.csv $OUT/sq/*counter_collection.csv $OUT/*/*kernel_trace.csv
ls -la $OUT
