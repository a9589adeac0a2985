#!/bin/bash
# usage (GPU box, through gpurun): tools/run_round_extras.sh <tag>   -> gpurun_out/<tag>/*
# beside tools/run_round_profiles.sh <tag>: config 4's one-launch search (trace, counters at 200 simulations, phase
# timers, the three act() routes side by side), the default trio beyond the listed instances (generic route, on-demand
# instances), root inference of the convolutional nets, the EZ nets, config 5's shape, the representation convolutions.
# Needs tools/bin/libmzsearch_prof.so (python tools/profile_search.py build, here) for the phase timers.
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
bash tools/rocprof_atari.sh 200 > $OUT/atari_trace.txt 2>&1
cd $REPO
bash tools/rocprof_tower_pmc.sh 200 > $OUT/tower_pmc.txt 2>&1
cd $REPO
python tools/profile_search.py run 2>&1 | grep -v amdgpu.ids > $OUT/search_phases.txt
python tools/bench_generic.py 2>&1 | grep -v amdgpu.ids > $OUT/generic.txt
python tools/bench_generic.py --round5 2>&1 | grep -v amdgpu.ids >> $OUT/generic.txt
python tools/bench_long.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_long.txt
python tools/bench_root_inference.py 2>&1 | grep -v amdgpu.ids > $OUT/root_inference.txt
python tools/bench_ez.py 2>&1 | grep -v amdgpu.ids > $OUT/ez_bench.txt
python tools/bench_ez.py 128 50 64 2>&1 | grep -v amdgpu.ids >> $OUT/ez_bench.txt
bash tools/rocprof_ez_root.sh 2>&1 | grep -v amdgpu.ids > $OUT/ez_root_trace.txt
cd $REPO
python tools/bench_cfg5.py 2>&1 | grep -v amdgpu.ids > $OUT/cfg5.txt
python tools/bench_atari.py 128 200 2>&1 | grep -v amdgpu.ids > $OUT/atari_bench.txt
python tools/bench_atari.py 1024 200 2>&1 | grep -v amdgpu.ids >> $OUT/atari_bench.txt
python tools/bench_repr_conv.py 2>&1 | grep -v amdgpu.ids > $OUT/repr_conv.txt
(python tools/stress_round5.py 400 2025; python tools/stress_round5.py 400 7; python tools/stress_round5.py 400 99; python tools/stress_parity.py 400 31; python tools/stress_parity.py 40 32 compact) 2>&1 | grep -v amdgpu.ids > $OUT/stress_round5.txt
bash tools/ab_fused.sh tools/bin/libmzsearch_split.so > $OUT/ab_split.log 2>&1
python tools/diag_stall.py 4000 0 0 2>&1 | grep -v amdgpu.ids > $OUT/stall.txt
python tools/diag_stall.py 4000 0 2048 2>&1 | grep -v amdgpu.ids >> $OUT/stall.txt
MUAX_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --cfg4-sims 200 --cfg4-acts 2 > $OUT/bench_2ranks_1gpu.json 2>> $OUT/bench.err
