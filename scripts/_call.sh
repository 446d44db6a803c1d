mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/c8_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c8_vq -- python $R/scripts/perf_vq.py 64 > $R/gpurun_out/c8_vq.log 2>&1
cd $R
f=$(find gpurun_out/c8_vq -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/c8_vq_kernel_stats.csv; rm -rf gpurun_out/c8_vq
timeout 300 python scripts/perf_vq.py 64 > gpurun_out/c8_vq_plain.log 2>&1
cat gpurun_out/c8_tests.log; grep "k_vq\|k_conv_few" gpurun_out/c8_vq_kernel_stats.csv | cut -c1-140; tail -n 2 gpurun_out/c8_vq_plain.log
