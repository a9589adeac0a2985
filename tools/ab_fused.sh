#!/bin/bash
# same-box A/B of fused-kernel builds (tools/bin/libmzsearch_*.so, same ABI): alternate the builds three times
for rep in 1 2 3; do
  for v in "" "$@"; do
    for w in cartpole lunarlander; do
      echo -n "rep $rep lib ${v:-product} $w: "
      MUAX_AMD_LIB=${v:+$PWD/$v} python bench.py --no-cpu-baseline --no-extras --workload $w 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['roofline']['kernel_ms'])"
    done
  done
done
