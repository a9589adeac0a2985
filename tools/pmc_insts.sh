#!/bin/bash
# usage: tools/pmc_insts.sh <tag> [variant-lib]   instruction-mix counters of the fused kernel (bench.py --no-extras, 5 steps)
TAG=$1; LIBV=$2
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmci_$TAG
mkdir -p $OUT
export MUAX_AMD_LIB=${LIBV:+$REPO/$LIBV}
B="python $REPO/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $B > $OUT/log1 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/pmc2 -o pmc2 -- $B > $OUT/log2 2>&1
python $REPO/tools/prof_summary.py $OUT | grep -v "^==" 
