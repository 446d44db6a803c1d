#!/bin/bash
# Run on the GPU box: the default bench line, then the same command under rocprofv3 (kernel trace + stats).
# usage: scripts/final_prof.sh <round tag, e.g. r02>   -> gpurun_out/<tag>_bench_line.json, gpurun_out/<tag>_prof/.../*kernel_stats.csv
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
tail -1 gpurun_out/${tag}_bench_line.json | cut -c1-600
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > $R/gpurun_out/${tag}_prof.json 2> $R/gpurun_out/${tag}_prof.err
cd $R
rm -f gpurun_out/${tag}_prof/*/*kernel_trace.csv; ls gpurun_out/${tag}_prof/*/ | head
