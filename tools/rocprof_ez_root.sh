#!/bin/bash
# kernel trace of the EZ nets' root inference (tools/bench_root_inference.py 128 ez_fused) -> gpurun_out/prof_ez_root/summary.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ez_root
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_root_inference.py 128 ez_fused > $OUT/bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_ez_root/trace/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# the last root inference of the run: from the last stem convolution on
idx = [i for i, r in enumerate(rows) if "16, 2>" in r[0] or "CIN" in r[0]]
start = max(0, len(rows) - 125)
for (n0, s0, e0), (n1, s1, e1) in list(zip(rows, rows[1:]))[start:]:
    print(f"{(s1 - e0) / 1e3:8.1f} {(e1 - s1) / 1e3:8.1f}  {n1[:110]}")
PY
rm -rf $OUT/trace
head -30 $OUT/summary.txt
