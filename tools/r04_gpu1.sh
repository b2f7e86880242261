cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/g1/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/g1/bench.err | tail -3) > gpurun_out/g1/bench.json
(timeout 900 python tools/strong_rehearsal.py --out gpurun_out/g1 2>&1 | tail -40) > gpurun_out/g1/rehearsal.log
(timeout 900 python tools/cli_wall.py --ranks 1 2 --out gpurun_out/g1/cli_wall.json 2>&1 | tail -20) > gpurun_out/g1/cli_wall.log
tail -5 gpurun_out/g1/pytest.log; cat gpurun_out/g1/rehearsal.log | tail -12; tail -c 1500 gpurun_out/g1/cli_wall.log
