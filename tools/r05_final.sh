#!/bin/bash
# final state of round 5 on the DEFAULT build: rocprof captures (gen / enc / misc), every GPU test, the bench line
#   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/r05_final.sh'
cd $GRAFT_REPO_ROOT
bash tools/capture_profiles.sh prof_gen gen > gpurun_out/prof_gen.log 2>&1
bash tools/capture_profiles.sh prof_enc enc > gpurun_out/prof_enc.log 2>&1
bash tools/capture_misc.sh prof_misc > gpurun_out/prof_misc.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_gen gpurun_out/prof_enc gpurun_out/prof_misc -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
mkdir -p gpurun_out/verify
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/verify/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/verify/bench.err | tail -1) > gpurun_out/verify/bench.json
(for i in 1 2; do python tools/encode_bench.py --iters 20 2>/dev/null; done) > gpurun_out/verify/encode.txt
tail -6 gpurun_out/verify/pytest.log; head -c 300 gpurun_out/verify/bench.json; echo; cat gpurun_out/verify/encode.txt
