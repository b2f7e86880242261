# PMC counters of HuBERT's conv1 under the conv2s128 variants (separate --pmc passes, kernel-trace only)
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06/pmc_s2
mkdir -p $OUT
for v in ${VARIANTS:-0 2 1}; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/v${v}_$i -o c -- python $R/tools/one_s2.py $v 31999 3 > $OUT/v${v}_$i.log 2>&1
    tail -1 $OUT/v${v}_$i.log
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/r06/pmc_s2/*/c_counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(d)):
        if "conv2s128" in r["Kernel_Name"] or "conv_mfma32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d.split("/")[3], {k: round(v / n[k] / 1e6, 3) for k, v in acc.items()}, "launches", max(n.values()) if n else 0)
PY
