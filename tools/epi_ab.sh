#!/bin/bash
# A/B of two builds of the library through the DISSC_HIP_LIB hook (how the conv_wino epilogue forms were compared, round 4):
#   hipcc ... -DDISSC_WINO_EPI2=0 -c dissc_amd/csrc/conv_wino.hip -o /tmp/cw0.o; hipcc -shared ... -o dissc_amd/libdissc_hip_epi0.so
#   gpurun -- 'bash tools/epi_ab.sh dissc_amd/libdissc_hip_epi0.so dissc_amd/libdissc_hip.so'
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for lib in "$@"; do
  DISSC_HIP_LIB=$PWD/$lib python bench.py --no-pipeline --no-split-bf16 --no-strong --no-d2h --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib bench', j['ms_per_step'])"
done
done
