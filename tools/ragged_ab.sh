cd $GRAFT_REPO_ROOT
for o in ragged_enum=1 ragged_enum=0; do echo "== $o"; DISSC_OPTIONS=$o python tools/ragged_cost.py 2>&1 | grep spread; DISSC_OPTIONS=$o python tools/encode_ragged.py 2>&1 | grep -v amdgpu | tail -4; done
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_hubert.py tests/test_gpu_edge_cases.py tests/test_gpu_predictors.py -x -q 2>&1 | tail -3
