# same-box A/B of DISSC_OPTIONS settings in the generator bench: bash tools/opt_ab.sh "" "key=v,key=v" ...
cd $GRAFT_REPO_ROOT
for rep in $(seq ${REPS:-2}); do
for o in "$@"; do
  DISSC_OPTIONS=$o python bench.py --steps 20 --no-cpu-baseline --no-pipeline --no-strong --no-split-bf16 --no-d2h 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('[$o]', 'ms_per_step', r['ms_per_step'])"
done; done
