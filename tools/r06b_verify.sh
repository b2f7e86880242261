#!/bin/bash
# last verification of the round: every GPU test, the bench line (with the committed PMC capture of this build), the encoder bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/verify
(timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | tail -25) > gpurun_out/verify/pytest.log 2>&1
(timeout 900 python bench.py 2> gpurun_out/verify/bench.err | tail -1) > gpurun_out/verify/bench.json
(for i in 1 2; do python tools/encode_bench.py --iters 40 2>/dev/null; done) > gpurun_out/verify/encode.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/verify/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/verify/smoke.txt
tail -3 gpurun_out/verify/pytest.log; head -c 400 gpurun_out/verify/bench.json; echo; cat gpurun_out/verify/encode.txt; tail -2 gpurun_out/verify/smoke.txt
