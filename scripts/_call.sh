mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for s in 0 4; do
  if [ $s = 0 ]; then unset WMAR_S_FC2; else export WMAR_S_FC2=$s; fi
  WMAR_ROOT=$R/build_alt/dev timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c13_$s -- python $R/scripts/perf_gpt.py 64 256 1 > $R/gpurun_out/c13_$s.log 2>&1
  f=$(find $R/gpurun_out/c13_$s -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/c13_${s}_kernel_stats.csv; rm -rf $R/gpurun_out/c13_$s
done
cd $R
for s in 0 4; do echo "== S_FC2=$s"; grep -v "^W2026\|^E2026" gpurun_out/c13_$s.log | tail -3; head -8 gpurun_out/c13_${s}_kernel_stats.csv | cut -c1-120; done
