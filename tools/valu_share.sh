#!/bin/bash
# VALU share of the generator / encoder kernels: instructions issued (VALU incl. MFMA, MFMA alone), VALU-active cycles, MFMA-busy cycles.
#   tools/valu_share.sh <out dir under gpurun_out> gen|enc     (table: tools/valu_share.py)
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1
WHAT=$2
mkdir -p "$OUT"
if [ "$WHAT" = gen ]; then
  CMD="python ${GRAFT_REPO_ROOT:-$PWD}/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-bf16 --no-pipeline --no-strong --no-d2h"
else
  CMD="python ${GRAFT_REPO_ROOT:-$PWD}/tools/encode_bench.py --iters 3"
fi
cd /tmp && export TMPDIR=/tmp
export DISSC_OPTIONS=multistream=0${EXTRA_OPTS:+,$EXTRA_OPTS}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$OUT/valu" -o valu -- $CMD > "$OUT/valu.log" 2>&1
echo "valu rc=$?"
find "$OUT" -type f ! -name "*counter_collection.csv" ! -name "*.log" -delete
