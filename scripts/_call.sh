mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sampler_bulk.py tests/test_gpu_watermark.py tests/test_gpu_gpt.py tests/test_gpu_gumbel.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c1_tests.log
timeout 300 python scripts/perf_sampler.py > gpurun_out/c1_perf_sampler.log 2>&1
AB_ROUNDS=3 AB_REPS=4 timeout 900 python scripts/ab_loop.py base tree > gpurun_out/c1_ab.log 2>&1
cat gpurun_out/c1_tests.log gpurun_out/c1_perf_sampler.log gpurun_out/c1_ab.log
