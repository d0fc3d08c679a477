#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/bench_short_pairs.py > gpurun_out/short_pairs.json 2> gpurun_out/short_pairs.err; tail -3 gpurun_out/short_pairs.err; python -c "
import json; [print(r) for r in json.load(open('gpurun_out/short_pairs.json'))]"
EDLIB_AMD_DEBUG=1 timeout 120 python tools/short_pairs_probe.py hw 2>&1 | sed -n '/==== second run/,$p' | head -20
EDLIB_AMD_DEBUG=1 timeout 120 python tools/short_pairs_probe.py nw 2>&1 | sed -n '/==== second run/,$p' | head -20
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
for cfg in 4 5; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-e2e 2> gpurun_out/quick_c$cfg.err | tee gpurun_out/quick_c$cfg.json | cut -c1-300
done
