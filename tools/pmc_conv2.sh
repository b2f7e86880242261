set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "256 2500 11 1" "256 2500 3 1" "64 40000 3 1" "64 40000 11 1"; do
  tag=$(echo $shape | tr " " "_")
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $R/gpurun_out/pmc2/$tag -o c -- python $R/tools/one_conv.py $shape 1 5 > $R/gpurun_out/pmc2_$tag.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc2/${tag}b -o c -- python $R/tools/one_conv.py $shape 1 5 >> $R/gpurun_out/pmc2_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc2/*/c_counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(d)):
        if "conv_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(d.split("/")[2], {k: round(v/1e6,2) for k,v in acc.items()})
PY
