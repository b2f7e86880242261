#!/bin/bash
# per-kernel times of the encoder: uniform vs padded rows
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/er -o er -- python $ROOT/tools/encode_ragged.py > /tmp/er.log 2>&1
grep -E "ms$" /tmp/er.log
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/er/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# three phases of 7 encodes each (2 warm-up + 5 timed); split the launches into thirds by count
n = len(rows) // 3
for ph in range(3):
    agg = collections.defaultdict(float)
    for r in rows[ph * n:(ph + 1) * n]:
        agg[r["Kernel_Name"][:70]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 7e3
    print("== phase", ph, "total %.1f us per encode" % sum(agg.values()))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:9]:
        print("  %9.1f us  %s" % (v, k))
PY
