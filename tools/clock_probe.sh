#!/bin/bash
# samples the shader clock / power while the generator bench runs (is the chip at its boost clock under sustained fp32 MFMA load?)
python bench.py --steps 1200 --warmup 5 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h > /tmp/b.json 2>/dev/null &
PID=$!
sleep 22
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo
  sleep 1
done
wait $PID
tail -c 400 /tmp/b.json | head -c 400; echo
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
