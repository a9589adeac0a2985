#!/bin/bash
# same-box A/B of recurrent-kernel builds (tools/bin/libmzsearch_*.so, same ABI): alternate the builds three times
for rep in 1 2 3; do
  for v in "" "$@"; do
    echo "== rep $rep lib ${v:-product}"
    MUAX_AMD_LIB=${v:+$PWD/$v} python tools/bench_tower.py 128 256 2>&1 | grep -v amdgpu.ids
  done
done
