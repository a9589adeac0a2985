#!/bin/bash
# The E = 32 network pass in isolation (tools/ubench_netpass_e32.hip) and the VALU / matrix-pipe co-issue probe
# (tools/ubench_coissue.hip): cycles + SQ counters -> gpurun_out/netpass_e32/.  Run through gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/netpass_e32
mkdir -p $OUT
$R/tools/bin/ubench_netpass_e32 200 512 > $OUT/cycles.txt 2>&1
$R/tools/bin/ubench_netpass_e32 200 256 >> $OUT/cycles.txt 2>&1
$R/tools/bin/ubench_netpass_e32_w1 200 256 >> $OUT/cycles.txt 2>&1
$R/tools/bin/ubench_coissue > $OUT/coissue.txt 2>&1
B="$R/tools/bin/ubench_netpass_e32 200 512"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $B > $OUT/pmc2.log 2>&1
python - <<PY
import glob, os, sqlite3
out = "$OUT"
lines = []
for db in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(db).cursor()
    lines.append(f"== {os.path.basename(db)} (rocprofv3 --pmc), per-dispatch averages ==")
    for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "where kernel_name like '%netpass%' group by kernel_name, counter_name"):
        lines.append(f"   {k[:30]:30s} {c:28s} {v:16.1f}  (n={n})")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
PY
find $OUT -name '*.db' -delete
cat $OUT/cycles.txt $OUT/coissue.txt $OUT/summary.txt
