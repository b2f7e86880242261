cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g3
(timeout 900 python -m pytest tests/test_gpu_pairw.py -x -q 2>&1 | tail -40) > gpurun_out/g3/pairw.log 2>&1
(timeout 600 python tools/pair_gate.py 2>&1 | tail -30) > gpurun_out/g3/gate.log
(timeout 900 python -m pytest tests/test_gpu_generator.py -x -q 2>&1 | tail -15) > gpurun_out/g3/gen.log 2>&1
(timeout 600 python bench.py --no-pipeline --no-split-bf16 2> gpurun_out/g3/bench.err | tail -3) > gpurun_out/g3/bench.json
cat gpurun_out/g3/pairw.log | tail -30; cat gpurun_out/g3/gate.log; tail -8 gpurun_out/g3/gen.log; head -c 400 gpurun_out/g3/bench.json
