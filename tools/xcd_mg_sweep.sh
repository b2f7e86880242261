cd $GRAFT_REPO_ROOT
for mg in 2 6; do
EXTRA_OPTS=xcd_order=1,xcd_mg=$mg bash tools/capture_profiles.sh prof_enc_mg$mg enc > gpurun_out/prof_enc_mg$mg.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_enc_mg$mg -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
done
