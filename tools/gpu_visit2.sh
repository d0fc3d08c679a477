#!/bin/bash
# GPU visit: parity suite, end-to-end one-shot rate, single-rank RCCL launch of bench.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 300 python tools/bench_e2e.py 2> gpurun_out/e2e.err | tee gpurun_out/e2e.json; tail -3 gpurun_out/e2e.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/rccl1.err | cut -c1-600 | tee gpurun_out/rccl1.json
tail -3 gpurun_out/rccl1.err
