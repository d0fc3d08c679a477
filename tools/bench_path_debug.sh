#!/bin/bash
# phase laps (EDLIB_AMD_DEBUG) of one LOC run and one PATH run on the read batch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for task in locations path; do
python - "$task" <<'PY' 2>&1 | grep -v scanGroup | tail -28
import sys, os
sys.path.insert(0, os.getcwd())
import edlib_amd
from edlib_amd import synth
T = synth.random_dna(12345, 5_000_000)
R = synth.illumina_reads(T, 262144)
b = edlib_amd.SharedBatch(R["reads"], T, mode="HW", task=sys.argv[1])
b.run()
os.environ["EDLIB_AMD_DEBUG"] = "1"
print("==== timed run", sys.argv[1], flush=True)
b.run()
PY
done
