#!/bin/bash
# usage (on the GPU box, through gpurun): tools/run_round_profiles.sh <tag>   -> gpurun_out/<tag>/*
# the measurements the docs quote: default bench line, rocprofv3 passes of both fused workloads, batch scaling,
# in-kernel phase timers (needs tools/bin/libmzsearch_prof.so: python tools/profile_phases.py build), API profile
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline --no-extras --workload lunarlander > $OUT/bench_lunarlander.json 2>> $OUT/bench.err
bash tools/rocprof_passes.sh ${TAG}_cartpole > $OUT/prof_cartpole.txt 2>&1
bash tools/rocprof_passes.sh ${TAG}_lunarlander --workload lunarlander > $OUT/prof_lunarlander.txt 2>&1
cd $REPO
for w in cartpole lunarlander; do
  for b in 1024 2048 4096 4112 6144 8192 16384 32768 65536; do
    python bench.py --no-cpu-baseline --no-extras --workload $w --roots $b 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$w', $b, l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])"
  done
done > $OUT/batch_scaling.txt
for w in cartpole lunarlander; do echo "## $w"; python tools/profile_phases.py run $w 2>&1 | grep -v amdgpu.ids; done > $OUT/phase_cycles.txt
python tools/profile_api.py 2>&1 | grep -v amdgpu.ids | head -60 > $OUT/api_profile.txt
