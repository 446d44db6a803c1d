mkdir -p gpurun_out
for nw in 2 1 4 2 1; do echo "NWA=$nw"; WMAR_CHAM_NWA=$nw timeout 600 python scripts/perf_cham.py 16 1024 0 2>&1 | grep "rep1"; done > gpurun_out/c10.log 2>&1
cat gpurun_out/c10.log
