#!/bin/bash
# tools/lin_traffic.sh OUT MG...: time + fabric traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the linears per sweep width
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/lin_traffic.py time "$@" > $OUT/time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python $R/tools/lin_traffic.py pmc "$@" > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python $R/tools/lin_traffic.py pmc "$@" > $OUT/write.log 2>&1
python $R/tools/lin_traffic.py tab $OUT "$@" > $OUT/traffic.txt 2>&1
cat $OUT/time.txt $OUT/traffic.txt
find $OUT -type f ! -name "*.txt" ! -name "*.log" -delete
