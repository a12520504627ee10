#!/bin/bash
# Timing-experiment build of the library: dense_fwd / dense_bwd issue 3 of every 8 MFMAs (-DBR_ABLATE_MFMA, WRONG results) - how much of
# each kernel's time is matrix-pipe time.  Run tools/mlp_bench.py with BR_LIB_PATH=tools/diag/libbinrec_ablate.so.  Never the product.
set -e
cd "$(dirname "$0")/../.."
python binary-recommendation_amd/build.py
B=binary-recommendation_amd/build
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I include -DBR_ABLATE_MFMA -x hip"
/opt/rocm/bin/hipcc $F -c binary-recommendation_amd/csrc/dense_fwd.hip -o /tmp/ablate_fwd.o &
/opt/rocm/bin/hipcc $F -c binary-recommendation_amd/csrc/dense_bwd.hip -o /tmp/ablate_bwd.o &
wait
OBJS=$(ls $B/*.o | grep -v "dense_fwd.hip.o\|dense_bwd.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diag/libbinrec_ablate.so $OBJS /tmp/ablate_fwd.o /tmp/ablate_bwd.o
ls -la tools/diag/libbinrec_ablate.so
