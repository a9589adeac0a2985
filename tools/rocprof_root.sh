cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_10
mkdir -p $OUT
python $GRAFT_REPO_ROOT/tools/bench_root_inference.py 2>&1 | grep -v amdgpu.ids | tee $OUT/root.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_root_inference.py 128 resnet_fused > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/trace/*.db
head -30 $OUT/summary.txt
