#!/bin/bash
# rocprofv3 kernel trace of the config-4-shaped run (tools/bench_atari.py) -> gpurun_out/prof_atari/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_atari
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_atari.py 128 ${1:-200} > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/gap_analysis.py $OUT > $OUT/gaps.txt 2>&1
rm -f $OUT/trace/*.db $OUT/trace/*/*.db
head -24 $OUT/summary.txt; cat $OUT/gaps.txt
