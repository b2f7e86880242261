#!/bin/bash
# One gpurun call that re-establishes the state of a build on the MI355X: every GPU test, the default bench line, the
# gate / knock-out records of respair_wino.hip.   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/gpu_verify.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/verify
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/verify/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/verify/bench.err | tail -1) > gpurun_out/verify/bench.json
(timeout 600 python tools/pair_gate.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/verify/pair_gate.txt
(timeout 600 python tools/pair_ko.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/verify/pair_ko.txt
tail -6 gpurun_out/verify/pytest.log; head -c 300 gpurun_out/verify/bench.json; echo; cat gpurun_out/verify/pair_ko.txt
