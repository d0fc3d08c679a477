#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_long_reads.py tests/test_gpu_soak.py -x -q 2>&1 | tail -3
timeout 60 python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import edlib_amd
from edlib_amd import synth
T = synth.random_dna(12345, 5_000_000)
out = {}
for m in (600, 768, 1024):
    R = synth.illumina_reads(T, 16384, m=m)["reads"]
    b = edlib_amd.SharedBatch(R, T, mode="HW", task="distance"); b.run(); st = b.run(); b.close()
    out[m] = round(st["run_ms"], 1)
print(json.dumps(out))
PY
