cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py > gpurun_out/r01_bench_final.json 2> gpurun_out/r01_bench_final.err
tail -1 gpurun_out/r01_bench_final.json | cut -c1-400
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_final -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_final.json 2> $R/gpurun_out/prof_bench_final.err
cd $R
rm -f gpurun_out/prof_bench_final/*/*kernel_trace.csv; ls gpurun_out/prof_bench_final/*/ | head
