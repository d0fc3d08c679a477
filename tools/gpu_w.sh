#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/bench_configs.py ${CFG_ARGS:-} 2>&1 | tee gpurun_out/configs.log
