#!/bin/bash
# A/B of the network pass (tools/ubench_netpass.hip) under rocprofv3: kernel trace + SQ counters.  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_netpass
mkdir -p $OUT
B="$REPO/tools/bin/ubench_netpass 200 256"
$B > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $B > $OUT/pmc2.log 2>&1
python - <<PY
import glob, os, sqlite3
out = "$OUT"
lines = [open(os.path.join(out, "plain.log")).read().rstrip()]
for db in sorted(glob.glob(os.path.join(out, "*", "*_results.db"))):
    cur = sqlite3.connect(db).cursor()
    name = os.path.basename(os.path.dirname(db))
    if name == "trace":
        lines.append("== rocprofv3 --kernel-trace --stats ==")
        for r in cur.execute("select name, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, avg(duration), count(*) from kernels group by name"):
            lines.append(f"   {r[0][:60]:60s} lds={r[1]} vgpr={r[2]} agpr={r[3]} sgpr={r[4]} avg_ns={r[5]:.0f} n={r[6]}")
    else:
        lines.append(f"== {name} (rocprofv3 --pmc), per-dispatch averages ==")
        try:
            for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
                lines.append(f"   {k[:44]:44s} {c:30s} {v:16.1f}  (n={n})")
        except Exception as e:
            lines.append(f"   ({e})")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
