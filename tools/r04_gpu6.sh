cd $GRAFT_REPO_ROOT
bash tools/capture_profiles.sh prof_gen gen > gpurun_out/prof_gen.log 2>&1
bash tools/capture_profiles.sh prof_enc enc > gpurun_out/prof_enc.log 2>&1
bash tools/capture_misc.sh prof_misc > gpurun_out/prof_misc.log 2>&1
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_pairw.py tests/test_gpu_generator.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -8) > gpurun_out/g6_tests.log 2>&1
tail -3 gpurun_out/prof_gen.log gpurun_out/prof_enc.log gpurun_out/prof_misc.log gpurun_out/g6_tests.log
# keep the merged output small: only the csv files the tables need
find gpurun_out/prof_gen gpurun_out/prof_enc gpurun_out/prof_misc -type f ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete
du -sh gpurun_out/prof_gen gpurun_out/prof_enc gpurun_out/prof_misc
