#!/bin/bash
# For the first box with more than one GPU: one command that exercises the N > 1 path on real hardware -- RCCL process group, key-table
# broadcast (harness.broadcast_key_table), per-step all_gather of codes / counts / p-values, barrier + max-over-ranks timing, one JSON
# line from rank 0.  (On CPU the same control flow is covered under gloo by tests/test_bench_distributed_cpu.py.)
# usage: scripts/smoke_2gpu.sh [n_gpus=2]
N=${1:-2}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus "$N" --steps 1 --warmup 1 --no-cpu-baseline --no-parity
