#!/bin/bash
# usage: tools/rocprof_passes.sh <tag> [extra bench args]   (run on the GPU box through gpurun)
TAG=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline $@"
BX="$B --no-extras"   # counter passes: the main loop only
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B > $OUT/bench_trace.log 2>&1   # the default run of bench.py (200 steps, 20 warm-up)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $BX --steps 5 --warmup 1 > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $BX --steps 5 --warmup 1 > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BX --steps 5 --warmup 1 > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $BX --steps 5 --warmup 1 > $OUT/bench_pmc4.log 2>&1
python $REPO/tools/prof_summary.py $OUT
rm -f $OUT/*/*.db $OUT/*/*/*.db   # (only the text summary travels back: gpurun_out/ is capped at 64 MiB)
