#!/bin/bash
# per-kernel time of the generator on a uniform and a ragged batch (serial launches): tools/ragged_prof.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export DISSC_OPTIONS=multistream=0
for w in ${RG_WIDTHS:-0 100}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rg$w -o rg -- python $ROOT/tools/ragged_cost.py $w > /tmp/rg$w.log 2>&1
  grep spread /tmp/rg$w.log
  python $ROOT/tools/kstats.py /tmp/rg$w 13 | head -${RG_LINES:-28}
done
