"""-m gpu: a fixed slice of the differential soak (tools/soak.py): random shared-target batches over every read-length
group of the lane-per-read kernels and random pair batches over the ring sizes, every field against the oracle.
(`python tools/soak.py 600 <seed>` is the open-ended form; profiles/README.md records the long runs.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [7, 8])
def test_soak_slice(engine, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "300", str(seed), "60"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["cases"] == 60 and not out["failures"]
