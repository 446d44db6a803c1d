#!/bin/bash
# usage: scripts/prof_top.sh <outdir-name> <python script + args...>   (run on the GPU box)
out=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$out -- python "$@" > /root/repo/gpurun_out/$out.log 2>&1
cd /root/repo
f=$(ls gpurun_out/$out/*/*kernel_stats.csv | head -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-72s %7s %12s %10.1f %6s" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"][:5]))
PY
