#!/bin/bash
# the EXPERIMENTAL build (DISSC_EXPERIMENTAL=1: kernels of failed gates compiled in) through every GPU test + its gate records
#   DISSC_EXPERIMENTAL=1 python -c "import __graft_entry__ as g; g.build()"; gpurun --timeout 3000 -- 'bash tools/r05_experimental.sh'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
python -c "
import ctypes, dissc_amd
v = ctypes.c_int(0); dissc_amd.lib.dissc_get_option(b'experimental', ctypes.byref(v)); print('experimental build:', v.value)" > gpurun_out/exp/pytest.log 2>&1
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) >> gpurun_out/exp/pytest.log 2>&1
(PYTHONPATH=. timeout 500 python tools/s2tc_gate.py 10 2>&1 | grep -v amdgpu.ids) > gpurun_out/exp/s2tc_gate.txt
(timeout 600 python tools/pair_gate.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/exp/pair_gate.txt
cat gpurun_out/exp/pytest.log; tail -12 gpurun_out/exp/s2tc_gate.txt; tail -5 gpurun_out/exp/pair_gate.txt
