#!/bin/bash
# first GPU contact: parity tests + a small reads-path timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/first_pytest.log
cat gpurun_out/first_pytest.log
python - <<'PY' 2>&1 | tee gpurun_out/first_bench.log
import time, numpy as np
import edlib_amd
from edlib_amd import synth
T = synth.random_dna(12345, 5_000_000)
for n in (4096, 65536, 262144):
    R = synth.illumina_reads(T, n)
    b = edlib_amd.SharedBatch(R["reads"], T, mode="HW", task="distance")
    b.run()
    t0 = time.time(); st = b.run(); dt = time.time() - t0
    print(n, "reads: wall %.3fs run_ms %.1f scan_ms %.1f launches %d GCUPS(wall) %.0f GCUPS(scan) %.0f word_steps %d ovf %d" % (
        dt, st["run_ms"], st["scan_ms"], st["scan_launches"], st["cells"]/dt/1e9, st["cells"]/st["scan_ms"]/1e6, st["word_steps"], st["overflow_units"]))
    arr = b.results_arrays()
    print("  ed hist", np.bincount(np.clip(arr["editDistance"], 0, 70))[:8], "max", arr["editDistance"].max())
    b.close()
PY
