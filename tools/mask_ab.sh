# same-box A/B of "wino8_mask" values (octal digits = stage classes C>=256 | C=128 | C=64; bits k11 k7 k3): bash tools/mask_ab.sh 390 391 ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in "$@"; do
  DISSC_OPTIONS=wino8_mask=$m python bench.py --steps 20 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('mask $m', 'ms_per_step', r['ms_per_step'], 'parity', r.get('parity', {}).get('rms'))"
done; done
