#!/bin/bash
# Diagnostic binaries (in-kernel stamps): built HERE with hipcc (cross-compiles without a GPU), run on the GPU box.
set -e
cd "$(dirname "$0")"
for t in fwd_stamps bwd_stamps; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ../../include -x hip $t.cpp -o $t.bin
done
ls -la *.bin
