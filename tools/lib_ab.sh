#!/bin/bash
# A/B of two builds of the library on one box: generator forward (bench.py headline loop) alternating DISSC_HIP_LIB
#   tools/lib_ab.sh [rounds]     (B = dissc_amd/libdissc_hip_alt.so)
R=${1:-3}
for r in $(seq 1 $R); do
  for lib in "" dissc_amd/libdissc_hip_alt.so; do
    DISSC_HIP_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h --no-latency 2>/dev/null \
      | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-default}', j['ms_per_step'], j['value'])"
  done
done
