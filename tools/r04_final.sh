#!/bin/bash
# final state of a round: rocprof captures (gen / enc / misc), every GPU test, the bench line, the gate records
cd $GRAFT_REPO_ROOT
bash tools/capture_profiles.sh prof_gen gen > gpurun_out/prof_gen.log 2>&1
bash tools/capture_profiles.sh prof_enc enc > gpurun_out/prof_enc.log 2>&1
bash tools/capture_misc.sh prof_misc > gpurun_out/prof_misc.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_gen gpurun_out/prof_enc gpurun_out/prof_misc -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
bash tools/gpu_verify.sh
