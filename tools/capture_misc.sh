#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the parts that have no per-layer table: the YAAPT tracker, the
# full pipeline leg of bench.py (encoder + predictors + generator + glue), the predictors alone.
#   tools/capture_misc.sh <out dir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o "$name" -- "$@" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run yaapt python $ROOT/tools/yaapt_bench.py
run pipeline python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-bf16 --no-strong --no-d2h --no-latency
find "$OUT" -name "*kernel_stats.csv"
