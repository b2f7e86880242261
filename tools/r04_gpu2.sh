cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > gpurun_out/g2/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/g2/bench.err | tail -3) > gpurun_out/g2/bench.json
(timeout 1200 python tools/strong_rehearsal.py --out gpurun_out/g2 2>&1 | tail -40) > gpurun_out/g2/rehearsal.log
(timeout 300 python tools/mall_probe.py 2>&1 | tail -20) > gpurun_out/g2/mall.log
(timeout 300 python tools/ragged_cost.py 2>&1 | tail -20) > gpurun_out/g2/ragged.log
tail -5 gpurun_out/g2/pytest.log; tail -12 gpurun_out/g2/rehearsal.log; cat gpurun_out/g2/mall.log gpurun_out/g2/ragged.log
