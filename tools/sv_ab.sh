#!/bin/bash
# A/B of the shared-transform form of conv_wino_kernel (option wino_sv) on single layers and the whole forward
python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
for o in ${SV_CONFIGS:-wino_sv=0 wino_sv=1 wino_sv=1,wino_cpr=16}; do
  echo "== $o"
  DISSC_OPTIONS=$o timeout 300 python tools/wino_gate.py time 2>&1 | grep -E "^C(256|128)" | cut -c1-110
  DISSC_OPTIONS=$o python bench.py --steps 10 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', j['value'], j['ms_per_step'])"
done
