#!/bin/bash
# (the trees build_dbg/wt_<sha> are `git archive <sha> cdmft-lanc-ed_amd include __graft_entry__.py | tar -x` extractions, each built with its own
#  __graft_entry__.build_engine(); build_dbg/ is scratch and git-ignored)
# round 4, item 1a: which commit slowed the real-vector Lanczos iteration (3.70 -> 4.00 ms)?  One process per extracted tree.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in c35fcfe a3bb891 32365a0 8ce257c d77ce39 fe2b908 c328e21 127a5f9; do
  timeout -k 10 200 python scripts/bisect_real.py build_dbg/wt_$d 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_bisect.log
timeout -k 10 200 python scripts/bisect_real.py . 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_bisect.log
timeout -k 10 200 python scripts/bisect_real.py . spread_banks=0 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_bisect.log
# item 1b: dispatch order of a group's blocks
for W in C5 C3 C4; do
  WORKLOAD=$W timeout -k 10 400 python scripts/ab.py "" "block_order=1" "block_order=2" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04_ab_block_order.log
