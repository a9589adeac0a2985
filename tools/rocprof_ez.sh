#!/bin/bash
# rocprofv3 kernel trace of the EZ-nets search (tools/bench_ez.py) -> gpurun_out/prof_ez/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ez
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_ez.py 128 50 ${1:-32} > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/trace/*.db $OUT/trace/*/*.db
head -16 $OUT/summary.txt; grep "mz::" $OUT/summary.txt | tail -12
