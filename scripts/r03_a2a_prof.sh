#!/bin/bash
# kernel-level breakdown of the two-transposes exchange with 4 thread ranks on one GPU (C3)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/a2a_prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT -o a2a -- python3 scripts/exchange_modes_c3.py 4 > gpurun_out/a2a_prof.log 2>&1
for f in $(find $OUT -name "*_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-230; done >> gpurun_out/a2a_prof.log
grep -v "^W2026\|^E2026" gpurun_out/a2a_prof.log | tail -60
