#!/bin/bash
# A/B of two builds of the engine on one box: ab_lib.sh <other libhxv.so> [WORKLOAD] [option sets...]
cd $GRAFT_REPO_ROOT
LIB=$1; shift; W=${1:-C3}; shift
WORKLOAD=$W timeout -k 10 400 python scripts/ab.py "$@" 2>&1 | grep -v amdgpu.ids
echo "--- $LIB"
HXV_LIB=$GRAFT_REPO_ROOT/$LIB WORKLOAD=$W timeout -k 10 400 python scripts/ab.py "$@" 2>&1 | grep -v amdgpu.ids
