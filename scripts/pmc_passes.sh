#!/bin/bash
# Run on the GPU box: separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over one decode role's replay.
# usage: scripts/pmc_passes.sh <role> [<role> ...]      -> gpurun_out/pmc/<role>_<counter>/...
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
for role in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/${role}_${ctr} -- python $GRAFT_REPO_ROOT/scripts/pmc_role.py $role > $GRAFT_REPO_ROOT/gpurun_out/pmc/${role}_${ctr}.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc -name "*counter_collection.csv" | head -20
