mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vq_paths.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/c11.log; cat gpurun_out/c11.log
