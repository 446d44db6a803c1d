#!/bin/bash
# usage: scripts/prof_top.sh <outdir-name> <python script (repo-relative) + args...>   (run on the GPU box)
out=$1; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
script=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$out -- python $R/$script "$@" > $R/gpurun_out/$out.log 2>&1
cd $R
f=$(ls gpurun_out/$out/*/*kernel_stats.csv | head -1)
rm -f gpurun_out/$out/*/*kernel_trace.csv
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-72s %7s %12s %10.1f %6s" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"][:5]))
PY
