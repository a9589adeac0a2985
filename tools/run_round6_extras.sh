#!/bin/bash
# usage (GPU box, through gpurun): tools/run_round6_extras.sh <tag>   -> gpurun_out/<tag>/*
# round 6's measurements beside tools/run_round_profiles.sh <tag>: config 4's one-launch search (trace, counters at 200
# simulations, phase timers with the tree step by phase, the LDS-tree A/B), the default trio beyond the listed instances
# (instances planned per policy, the generic route undivided / in chunks), a synchronised act() taken apart, root
# inference of the convolutional nets, the EZ nets, config 5's shape, randomised parity campaigns, the 2-rank dry run.
# Needs tools/bin/libmzsearch_prof.so (python tools/profile_search.py build, here) for the phase timers.  Every command
# under `timeout`: a hung kernel must cost minutes, not the call's whole limit.
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 400 bash tools/rocprof_atari.sh 200 > $OUT/atari_trace.txt 2>&1
cd $REPO
timeout 600 bash tools/rocprof_tower_pmc.sh 200 > $OUT/tower_pmc.txt 2>&1
cd $REPO
timeout 300 python tools/profile_search.py run 2>&1 | grep -v amdgpu.ids > $OUT/search_phases.txt
# (tools/bin/libmzsearch_prof_heads.so: MZ_PROF_HEADS=1 python tools/profile_search.py build)
MZ_PROF_HEADS=1 timeout 300 python tools/profile_search.py run 2>&1 | grep -v amdgpu.ids > $OUT/search_phases_heads.txt
for r in 1 2 3; do for v in 1 0; do echo -n "rep $r MZS_SEARCH_LDS_TREE=$v: "; MZS_SEARCH_LDS_TREE=$v timeout 120 python tools/bench_atari.py 128 200 2>&1 | grep -v amdgpu.ids | tail -1; done; done > $OUT/ab_ldstree.txt 2>&1
timeout 300 python tools/bench_generic.py --round6 2>&1 | grep -v amdgpu.ids > $OUT/generic.txt
timeout 300 python tools/bench_generic.py 2>&1 | grep -v amdgpu.ids >> $OUT/generic.txt
timeout 300 python tools/bench_long.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_long.txt
for r in 1 2 3; do timeout 120 python tools/sync_gap.py 2>&1 | grep -v amdgpu.ids; done > $OUT/sync_gap.txt
timeout 300 python tools/bench_root_inference.py 2>&1 | grep -v amdgpu.ids > $OUT/root_inference.txt
timeout 300 python tools/bench_ez.py 2>&1 | grep -v amdgpu.ids > $OUT/ez_bench.txt
timeout 300 python tools/bench_ez.py 128 50 64 2>&1 | grep -v amdgpu.ids >> $OUT/ez_bench.txt
cd $REPO
timeout 300 python tools/bench_cfg5.py 2>&1 | grep -v amdgpu.ids > $OUT/cfg5.txt
timeout 300 python tools/bench_atari.py 128 200 2>&1 | grep -v amdgpu.ids > $OUT/atari_bench.txt
timeout 300 python tools/bench_atari.py 1024 200 2>&1 | grep -v amdgpu.ids >> $OUT/atari_bench.txt
(timeout 400 python tools/stress_round5.py 400 2026; timeout 400 python tools/stress_round5.py 400 8; timeout 300 python tools/stress_parity.py 400 33; timeout 300 python tools/stress_parity.py 40 34 compact; timeout 400 python tools/stress_stepwise.py 300 35) 2>&1 | grep -v amdgpu.ids > $OUT/stress_round5.txt
MUAX_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cfg4-sims 200 --cfg4-acts 2 > $OUT/bench_2ranks_1gpu.json 2>> $OUT/bench.err
