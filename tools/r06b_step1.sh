#!/bin/bash
# round 6 (second session), step 1: attention in XCD order (xcd_order bit 3) + sweep-width sweep of the linears
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hubert.py -x -q -m gpu > $O/pytest_hubert.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_hubert.log
for rep in 1 2 3; do for o in 3 11; do
  echo -n "[enc xcd_order=$o] "; DISSC_OPTIONS=xcd_order=$o python tools/encode_bench.py --iters 20 2>/dev/null | tail -1
done; done > $O/attn_ab.txt 2>&1
cat $O/attn_ab.txt
bash tools/lin_traffic.sh r06b1/lin 0 1 2 3 6 12 24 > $O/lin_traffic.txt 2>&1
cat $O/lin_traffic.txt
# attention traffic under both orders (FETCH_SIZE / WRITE_SIZE, separate passes, serial launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for o in 3 11; do for c in FETCH_SIZE WRITE_SIZE; do
  DISSC_OPTIONS=multistream=0,hubert_split=0,xcd_order=$o timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/attn_${o}_$c -o a -- python $R/tools/encode_bench.py --iters 3 > $R/$O/attn_${o}_$c.log 2>&1
done; done
cd $R
python - <<'PY' | tee gpurun_out/r06b1/attn_traffic.txt
import csv, glob, collections
for o in (3, 11):
    tot = {}
    for c, sc in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        per = collections.defaultdict(float)
        for f in glob.glob(f"gpurun_out/r06b1/attn_{o}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "attn_fused" in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"]) * sc
        v = sorted(per.values())
        tot[c] = v[len(v) // 2] / 1e9 if v else float("nan")
    print(f"xcd_order={o}: attention fabric traffic per launch: read {tot['FETCH_SIZE']:.3f} + write {tot['WRITE_SIZE']:.3f} = {tot['FETCH_SIZE'] + tot['WRITE_SIZE']:.3f} GB (algorithmic 0.196)")
PY
find $O -type f ! -name "*.txt" ! -name "*.log" -delete
