cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ez2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_ez.py 128 50 32 > $OUT/bench.log 2>&1
python - <<'PY'
import glob, sqlite3, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_ez2/trace/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "ez_recurrent" in r[0]]
mid = idx[len(idx) * 3 // 4]
for (n0, s0, e0), (n1, s1, e1) in list(zip(rows, rows[1:]))[mid - 1:mid + 14]:
    print(f"{(s1 - e0) / 1e3:8.1f} {(e1 - s1) / 1e3:8.1f}  {n1[:100]}")
PY
rm -rf $OUT/trace
