#!/bin/bash
# builds and runs the matrix-pipe micro-benchmark on the GPU box
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_mix mfma_mix.hip
/tmp/mfma_mix
