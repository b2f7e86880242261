cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g5
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > gpurun_out/g5/pytest.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/g5/bench.err | tail -3) > gpurun_out/g5/bench.json
(DISSC_OPTIONS=pair_wino=0 timeout 600 python bench.py --no-pipeline --no-split-bf16 --no-strong --no-cpu-baseline --no-d2h 2>/dev/null | tail -1) > gpurun_out/g5/bench_pw0.json
(DISSC_OPTIONS=pair_wino=2 timeout 600 python bench.py --no-pipeline --no-split-bf16 --no-strong --no-cpu-baseline --no-d2h 2>/dev/null | tail -1) > gpurun_out/g5/bench_pw2.json
(timeout 1200 python tools/strong_rehearsal.py --out gpurun_out/g5 2>&1 | tail -40) > gpurun_out/g5/rehearsal.log
tail -5 gpurun_out/g5/pytest.log; tail -12 gpurun_out/g5/rehearsal.log; for f in bench bench_pw0 bench_pw2; do python -c "
import json,sys
j=json.loads(open('gpurun_out/g5/$f.json').read().strip().splitlines()[-1]); print('$f', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['algorithmic_frac'], j.get('parity',{}).get('rms'))"; done
