#!/bin/bash
# rocprofv3 kernel trace of tools/bench_stepwise.py -> gpurun_out/prof_stepwise/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_stepwise
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_stepwise.py > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/trace/*.db
head -30 $OUT/summary.txt
