#!/bin/bash
# MFMA utilisation counters of the ResNet recurrent kernel (separate --pmc pass, no traces)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tower_pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_atari.py 128 ${1:-200} > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/*/*.db $OUT/*/*/*.db
echo "# counters of a ${1:-200}-simulation search per launch (tools/rocprof_tower_pmc.sh ${1:-200}): compare with that many simulations' time"
grep "tower\|search" $OUT/summary.txt
