mkdir -p gpurun_out
bash scripts/pmc_final.sh > gpurun_out/c12_pmc.log 2>&1
tail -12 gpurun_out/c12_pmc.log
