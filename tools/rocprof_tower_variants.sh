#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of the fused recurrent kernel for experiment builds of the
# library given as tools/bin/libmz_<NAME>.so: bash tools/rocprof_tower_variants.sh NAME...   ("main" = the shipped one)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/muax_amd/lib/libmzsearch.so /tmp/keep.so
for v in "$@"; do
  [ "$v" != main ] && cp $R/tools/bin/libmz_$v.so $R/muax_amd/lib/libmzsearch.so
  OUT=$R/gpurun_out/prof_var_$v
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/bench_tower.py 128 > $OUT/bench.log 2>&1
  python $R/tools/prof_summary.py $OUT > /dev/null 2>&1
  rm -f $OUT/trace/*.db
  echo "== $v"; grep "tower" $OUT/summary.txt | grep avg_ns
  cp /tmp/keep.so $R/muax_amd/lib/libmzsearch.so
done
