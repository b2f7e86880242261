#!/bin/bash
# A/B of conv_wino_kernel options on single layers and the whole forward:  SV_CONFIGS="wino_sv=0 wino_sv=1" SV_ROWS="256|128"
python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
for o in ${SV_CONFIGS:-wino_sv=0 wino_sv=1}; do
  echo "== $o"
  DISSC_OPTIONS=$o timeout 300 python tools/wino_gate.py time 2>&1 | grep -E "^C(${SV_ROWS:-256|128|64})" | cut -c1-110
  DISSC_OPTIONS=$o python bench.py --steps 10 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', j['value'], j['ms_per_step'])"
done
