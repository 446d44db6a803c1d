#!/bin/bash
# Round 6, on the GPU box: the bench line + its rocprof summary, the small-batch loops, RAR-XL and Chameleon-7B under rocprofv3, and the
# PMC passes of the small-batch kernels.  -> gpurun_out/r06_*
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
bash scripts/final_prof.sh r06 > gpurun_out/r06_final_prof.log 2>&1
f=$(find gpurun_out/r06_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_bench_kernel_stats.csv
cd /tmp; export TMPDIR=/tmp
prof() {   # tag, command...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/_p_$tag -- "$@" > $R/gpurun_out/${tag}.log 2>&1
  f=$(find $R/gpurun_out/_p_$tag -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${tag}_kernel_stats.csv; rm -rf $R/gpurun_out/_p_$tag
}
prof r06_small_batch_b1 python $R/scripts/perf_gpt.py 1 256 1
prof r06_small_batch_b5 python $R/scripts/perf_gpt.py 5 256 1
prof r06_rar_xl_b64 python $R/scripts/perf_rar.py 64 1
prof r06_chameleon7b_b16_1024tok python $R/scripts/perf_cham.py 16 1024 0
mkdir -p $R/gpurun_out/pmc_small
for role in qkv attn proj fc1 fc2; do for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_small/${role}_${ctr} -- python $R/scripts/pmc_role.py $role 5 > $R/gpurun_out/pmc_small/${role}_${ctr}.log 2>&1
done; done
cd $R
python scripts/pmc_small.py
rm -rf gpurun_out/pmc_small gpurun_out/r06_prof
grep -E "steps=|ms/step|img/s" gpurun_out/r06_small_batch_b1.log gpurun_out/r06_small_batch_b5.log gpurun_out/r06_rar_xl_b64.log gpurun_out/r06_chameleon7b_b16_1024tok.log | tail -12
tail -1 gpurun_out/r06_bench_line.json | cut -c1-300
