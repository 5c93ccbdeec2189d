#!/bin/bash
# same-box A/B of variant BUILDS (scripts/build_variant.py -> gpurun_ab/libhxv_<tag>.so): one process per library, the shipped one first and last
# usage: lib_ab.sh <tag> [<tag> ...]      (WORKLOAD=C3|C4|C5 as for scripts/ab.py)
for T in base "$@" base; do
  if [ "$T" = "base" ]; then L=cdmft-lanc-ed_amd/lib/libhxv.so; else L=gpurun_ab/libhxv_$T.so; fi
  echo "== $T"
  HXV_LIB=$PWD/$L python scripts/ab.py '' 2>&1 | grep "^C[2-5]"
done
