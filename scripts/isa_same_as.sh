#!/bin/bash
# Is the DEVICE code of the working tree the one of <commit>?  Compiles every .hip source of both trees to gfx950 assembly (device only) and
# diffs them ignoring comments, debug directives and the per-compilation cuid symbol.  Used to show that the kernels shipped at the end of a round
# are the ones its profiles were collected on (round 6: profiles of 6f0e33e).      usage: scripts/isa_same_as.sh <commit>
C=${1:?commit}; T=$(mktemp -d)
mkdir -p $T/old $T/new
git archive $C cdmft-lanc-ed_amd/csrc include | tar -x -C $T/old
f() { grep -v "^\s*;\|\.file\|\.ident\|^\s*\.loc\|debug\|__hip_cuid" $1; }
rc=0
for S in cdmft-lanc-ed_amd/csrc/*.hip; do
  B=$(basename $S .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S -o $T/new/$B.s $S 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S -o $T/old/$B.s $T/old/$S 2>/dev/null
  n=$(diff <(f $T/old/$B.s) <(f $T/new/$B.s) | wc -l)
  echo "$B: $n differing lines"
  [ "$n" = "0" ] || rc=1
done
rm -rf $T
exit $rc
