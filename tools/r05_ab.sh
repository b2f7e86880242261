#!/bin/bash
# round 5, verdict task 4: the dilated conv1 instances of the C >= 128 stages as F(6,3) instead of F(5,4) where the tile table
# (profiles/r05/dilated_twins.md) says the grid quantises better -- same-box forward A/B, 2 repetitions.  + the node CPU leg.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
REPS=2 bash tools/opt_ab.sh "" "wino8_r4_mask=119795720" "wino8_r4_mask=119771144" "wino8_r4_mask=52531208" "wino8_r4_mask=65138696" > gpurun_out/r5c/dilated_ab.txt 2>&1
(echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "nproc: $(nproc)"; python - <<'PY'
import bench, json
print("quota", bench.cgroup_cpu_limit(), "sets16", len(bench._physical_core_sets(16)), "sets32", len(bench._physical_core_sets(32)))
for th in (16, 32):
    print(th, json.dumps(bench.cpu_baseline_node(th, 32, 500, duration=5.0, lead=15.0)))
PY
) > gpurun_out/r5c/cpu_node.txt 2>&1
cat gpurun_out/r5c/dilated_ab.txt gpurun_out/r5c/cpu_node.txt
