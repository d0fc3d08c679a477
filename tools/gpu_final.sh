#!/bin/bash
# last GPU visit of a round: tests, smoke, the three bench lines, and the small tables (latency, LOC / PATH cost,
# short pairs, read lengths)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/smoke.log
for cfg in 2 4 5; do
  timeout 900 python bench.py --config $cfg 2> gpurun_out/bench_c$cfg.err > gpurun_out/final_c$cfg.json
  python -c "
import json; d = json.load(open('gpurun_out/final_c$cfg.json')); print($cfg, d['value'], d['ms_per_step'], d.get('parity_sample', {}).get('bit_exact'), d.get('e2e', {}).get('value'))"
done
timeout 200 build/latency edlib_amd/libedlib.so > gpurun_out/final_latency.json 2>/dev/null; cat gpurun_out/final_latency.json
timeout 300 python tools/bench_path.py 262144 2>/dev/null | tee gpurun_out/final_path.json
timeout 300 python tools/bench_short_pairs.py 2>/dev/null > gpurun_out/final_short_pairs.json; python -c "
import json; [print(r) for r in json.load(open('gpurun_out/final_short_pairs.json'))]"
