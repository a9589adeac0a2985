#!/bin/bash
# Counters of config 4's dominant kernel (mz_resnet_search_kernel, one launch per act) in separate --pmc passes (no
# traces), every pass at ${1:-200} simulations per launch -- the count the bench line's kernel time refers to:
#   pmc1  matrix-pipe counters (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES ...)
#   pmc2  FETCH_SIZE   pmc3  WRITE_SIZE   (HBM side; KB; gfx950: reads x 2, /opt/skills/guides/MI355X_MICROARCH.md "HBM")
#   pmc4  L2 requests / hits / misses (the convolution weights every workgroup re-streams from L2 per pass)
# usage (GPU box, through gpurun): tools/rocprof_tower_pmc.sh [simulations]
S=${1:-200}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tower_pmc
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/tools/bench_atari.py 128 $S"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- $B > $OUT/bench1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc2 -- $B > $OUT/bench2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc3 -o pmc3 -- $B > $OUT/bench3.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $B > $OUT/bench4.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
rm -f $OUT/*/*.db $OUT/*/*/*.db
echo "# counters of a $S-simulation search, per launch of mz_resnet_search_kernel (tools/rocprof_tower_pmc.sh $S): compare with $S simulations' time"
grep "tower\|search\|^==" $OUT/summary.txt
