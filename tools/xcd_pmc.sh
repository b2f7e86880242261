# PMC traffic of the encoder under the XCD-aware workgroup order (tools/xcd_ab.sh has the timings)
cd $GRAFT_REPO_ROOT
EXTRA_OPTS=xcd_order=3 bash tools/capture_profiles.sh prof_enc_xcd enc > gpurun_out/prof_enc_xcd.log 2>&1
cd $GRAFT_REPO_ROOT

cd $GRAFT_REPO_ROOT
find gpurun_out/prof_enc_xcd -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
cat gpurun_out/prof_enc_xcd.log.log | grep rc=
for rep in 1 2 3; do for o in 0 3; do
  echo -n "[enc xcd_order=$o] "; DISSC_OPTIONS=xcd_order=$o python tools/encode_bench.py --iters 10 2>/dev/null | tail -1
done; done
