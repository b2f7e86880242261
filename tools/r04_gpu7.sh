cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/g7/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/g7/bench.err | tail -1) > gpurun_out/g7/bench.json
(timeout 600 python tools/pair_gate.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/g7/pair_gate.txt
(timeout 600 python tools/pair_ko.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/g7/pair_ko.txt
tail -6 gpurun_out/g7/pytest.log; head -c 300 gpurun_out/g7/bench.json; echo; cat gpurun_out/g7/pair_ko.txt
