set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "256 2500 11 1" "256 2500 3 1" "768 500 1 1" "64 40000 3 1"; do
  tag=$(echo $shape | tr " " "_")
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_conv/$tag -o c -- python $R/tools/one_conv.py $shape 1 5 > $R/gpurun_out/pmc_conv_$tag.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_conv/${tag}b -o c -- python $R/tools/one_conv.py $shape 1 5 >> $R/gpurun_out/pmc_conv_$tag.log 2>&1
  tail -1 $R/gpurun_out/pmc_conv_$tag.log
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_conv/*/c_counter_collection.csv")):
    acc=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(d)):
        if "conv_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(d.split("/")[2], {k: round(v/1e6,2) for k,v in acc.items()})
PY
