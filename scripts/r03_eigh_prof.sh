#!/bin/bash
# kernel-level breakdown of hxv_eigh_lowest at C3 (2 states, ncv = 20)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/eigh_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o eigh -- python3 scripts/eigh_c3.py > gpurun_out/eigh_prof.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' >> gpurun_out/eigh_prof.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.3f s" % (tot / 1e9))
for r in rows[:22]:
    print("%-90s calls %6s  avg %9.3f ms  total %7.3f s  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e9, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -40 gpurun_out/eigh_prof.log
