#!/bin/bash
# attention kernel rewrite (attn.hip): tests, encoder timing, per-kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hubert.py tests/test_gpu_kmeans.py -x -q -m gpu > $O/pytest_hubert.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_hubert.log; tail -5 $O/pytest_hubert.log
for rep in 1 2 3; do python tools/encode_bench.py --iters 20 2>/dev/null | tail -1; done > $O/encode.txt; cat $O/encode.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
DISSC_OPTIONS=multistream=0,hubert_split=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o t -- python $R/tools/encode_bench.py --iters 5 > $R/$O/trace.log 2>&1
cd $R
grep -h "attn_fused\|lin128\|conv2s128" $(find $O/trace -name "*kernel_stats.csv") | cut -c1-200 | tee $O/kstats.txt
find $O -type f ! -name "*.txt" ! -name "*.log" -delete
