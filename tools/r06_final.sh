#!/bin/bash
# final state of round 6 on the DEFAULT build: rocprof captures (gen / enc / misc; serial-launch PMC passes + the benched schedule's
# kernel trace), every GPU test, the bench line, the encoder bench, the gates of the round's kernels
#   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/r06_final.sh'
cd $GRAFT_REPO_ROOT
bash tools/capture_profiles.sh prof_gen gen > gpurun_out/prof_gen.log 2>&1
bash tools/capture_profiles.sh prof_enc enc > gpurun_out/prof_enc.log 2>&1
bash tools/capture_misc.sh prof_misc > gpurun_out/prof_misc.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_gen gpurun_out/prof_enc gpurun_out/prof_misc -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
mkdir -p gpurun_out/verify
(timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | tail -25) > gpurun_out/verify/pytest.log 2>&1
(timeout 900 python bench.py 2> gpurun_out/verify/bench.err | tail -1) > gpurun_out/verify/bench.json
(for i in 1 2; do python tools/encode_bench.py --iters 40 2>/dev/null; done) > gpurun_out/verify/encode.txt
(timeout 600 python tools/lin128_gate.py 0 1 2>&1 | grep -v amdgpu.ids) > gpurun_out/verify/lin128_gate.txt
(timeout 600 python tools/conv2s128_gate.py 0 1 3 2>&1 | grep -v amdgpu.ids) > gpurun_out/verify/conv2s128_gate.txt
(BATCHES=1,2,4,8,16,32,64 python tools/batch_scaling.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/verify/batch_scaling.txt
tail -4 gpurun_out/verify/pytest.log; head -c 300 gpurun_out/verify/bench.json; echo; cat gpurun_out/verify/encode.txt; tail -7 gpurun_out/verify/lin128_gate.txt; tail -6 gpurun_out/verify/conv2s128_gate.txt; cat gpurun_out/verify/batch_scaling.txt
