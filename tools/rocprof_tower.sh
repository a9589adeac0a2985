#!/bin/bash
# rocprofv3 kernel durations of the fused recurrent kernel (both launch shapes) under tools/bench_tower.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tower
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/bench_tower.py ${1:-128} > $OUT/bench.log 2>&1
python $R/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/trace/*.db
grep "tower" $OUT/summary.txt | grep avg_ns; tail -1 $OUT/bench.log
