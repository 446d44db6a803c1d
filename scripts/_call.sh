mkdir -p gpurun_out
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/c16_smoke.log 2>&1; tail -3 gpurun_out/c16_smoke.log
