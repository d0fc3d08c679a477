#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for hb in "" 1; do
  EDLIB_AMD_NOHOLDBACK=$hb timeout 600 python bench.py --config 4 --no-cpu-baseline --no-e2e --steps 5 2> gpurun_out/quick_c4.err | tee gpurun_out/quick_c4_hb$hb.json | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('noholdback=$hb', d['ms_per_step'], d['value'], d['roofline']['scan_ms_per_step'], d['roofline']['scan_launches_per_step'], d.get('parity'))"
done
timeout 300 python tools/bench_short_pairs.py 2> gpurun_out/short_pairs.err | tee gpurun_out/short_pairs.json | python -c "
import sys, json; [print(r) for r in json.load(sys.stdin)]"
timeout 900 python -m pytest tests/test_gpu_rings.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
