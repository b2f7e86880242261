cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
(timeout 900 python -m pytest tests/test_gpu_pairw.py -x -q 2>&1 | tail -15) > gpurun_out/g4/pairw.log 2>&1
(timeout 600 python tools/pair_gate.py 2>&1 | tail -30) > gpurun_out/g4/gate.log
cat gpurun_out/g4/pairw.log | tail -12; cat gpurun_out/g4/gate.log
