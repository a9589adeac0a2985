#!/bin/bash
# rocprofv3 kernel trace of the training-step bench -> gpurun_out/prof_train/ + text summary
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_train.py > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
tail -5 $OUT/bench.log; head -20 $OUT/summary.txt
