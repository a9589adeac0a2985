#!/bin/bash
# round-4 additions to tools/run_round_profiles.sh <tag>: config 4's one-launch search (trace, MFMA counters, phase
# timers, the three act() routes side by side), the generic / on-demand routes of the default trio
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
bash tools/rocprof_atari.sh 200 > $OUT/atari_trace.txt 2>&1
cd $REPO
bash tools/rocprof_tower_pmc.sh > $OUT/tower_pmc.txt 2>&1
cd $REPO
python tools/profile_search.py run 2>&1 | grep -v amdgpu.ids > $OUT/search_phases.txt
python tools/bench_generic.py 2>&1 | grep -v amdgpu.ids > $OUT/generic.txt
python tools/bench_root_inference.py 2>&1 | grep -v amdgpu.ids > $OUT/root_inference.txt
python tools/bench_ez.py 2>&1 | grep -v amdgpu.ids > $OUT/ez_bench.txt
python tools/bench_cfg5.py 2>&1 | grep -v amdgpu.ids > $OUT/cfg5.txt
python tools/bench_atari.py 128 200 2>&1 | grep -v amdgpu.ids > $OUT/atari_bench.txt
python tools/bench_atari.py 1024 200 2>&1 | grep -v amdgpu.ids >> $OUT/atari_bench.txt
