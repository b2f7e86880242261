#!/bin/bash
# rocprofv3 captures for profiles/rNN (run on the GPU box through gpurun):
#   tools/capture_profiles.sh <out dir under gpurun_out> gen|enc
# One --kernel-trace --stats pass and three --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE: separate passes,
# never combined with other trace domains), all with serial launches (multistream=0) so that per-kernel
# durations do not overlap.  Tables: tools/prof_tables.py.
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/$1
WHAT=$2
mkdir -p "$OUT"
if [ "$WHAT" = gen ]; then
  CMD="python ${GRAFT_REPO_ROOT:-$PWD}/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-split-bf16 --no-pipeline --no-strong --no-d2h --no-latency"
else
  CMD="python ${GRAFT_REPO_ROOT:-$PWD}/tools/encode_bench.py --iters 8"
fi
cd /tmp && export TMPDIR=/tmp
export DISSC_OPTIONS=multistream=0${EXTRA_OPTS:+,$EXTRA_OPTS}
run() {  # name, rocprof args...
  local name=$1; shift
  timeout 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $CMD > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run trace --kernel-trace --stats
# the BENCHED schedule (multistream=1: parallel streams), kernel trace only: its wall span per forward closes the check "kernel time
# per step <= ms_per_step" from files alone (tools/trace_span.py; round 5 verdict, item 5)
DISSC_OPTIONS=${EXTRA_OPTS:-} run trace_ms --kernel-trace
export DISSC_OPTIONS=multistream=0${EXTRA_OPTS:+,$EXTRA_OPTS}
run sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
find "$OUT" -name "*.csv" | head -20
