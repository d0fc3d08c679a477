#!/bin/bash
# phase laps (EDLIB_AMD_DEBUG) of single edlibAlign() calls: 100 x 100 NW distance and path, steady state
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["EDLIB_AMD_DEBUG"] = "1"
import edlib_amd
q = bytes(b"ACGT"[(i * 7 + i // 3 + (i % 17 == 0)) & 3] for i in range(100))
t = bytes(b"ACGT"[(i * 7 + i // 3) & 3] for i in range(100))
for task in ("distance", "path"):
    for i in range(6):
        sys.stderr.write("---- %s call %d\n" % (task, i)); sys.stderr.flush()
        edlib_amd.align_raw(q, t, "NW", task, -1)
PY
