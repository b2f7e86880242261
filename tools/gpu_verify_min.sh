cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/verify
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/verify/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/verify/bench.err | tail -1) > gpurun_out/verify/bench.json
tail -6 gpurun_out/verify/pytest.log; head -c 400 gpurun_out/verify/bench.json; echo
