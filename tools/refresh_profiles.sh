#!/bin/bash
# After `gpurun -- bash tools/r04_final.sh`: turn the merged captures under gpurun_out/ into the tracked files of profiles/<round>.
#   bash tools/refresh_profiles.sh [r04]
set -e
cd "$(dirname "$0")/.."
R=${1:-r04}
G=gpurun_out
for w in gen enc; do
  python tools/prof_tables.py $w --trace $G/prof_$w/trace/trace_kernel_trace.csv --sq $G/prof_$w/sq/sq_counter_collection.csv \
    --fetch $G/prof_$w/fetch/fetch_counter_collection.csv --write $G/prof_$w/write/write_counter_collection.csv \
    --md profiles/$R/${w}_per_layer.md $([ $w = gen ] && echo --json profiles/$R/hbm_traffic.json) > /dev/null
  cp $G/prof_$w/trace/trace_kernel_stats.csv profiles/$R/${w}_kernel_stats.csv
done
cp $G/prof_misc/pipeline/pipeline_kernel_stats.csv profiles/$R/pipeline_kernel_stats.csv
cp $G/prof_misc/yaapt/yaapt_kernel_stats.csv profiles/$R/yaapt_kernel_stats.csv
for f in pair_gate.txt pair_ko.txt; do [ -f $G/verify/$f ] && cp $G/verify/$f profiles/$R/$f; done
[ -s $G/verify/bench.json ] && cp $G/verify/bench.json profiles/$R/bench.json
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
h = json.load(open("profiles/$R/hbm_traffic.json"))
print("hbm_traffic.json hash", h.get("kernel_source_hash"), "sources now", bench.kernel_source_hash())
PY
