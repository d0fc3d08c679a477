"""BASELINE config 2 at FULL size, every read: the engine's resident results for bench.py's 1,000,000 x 150 bp HW batch against
the unmodified reference's answers for the same batch (tests/golden/c2_full_ref.npz: tools/full_parity_c2.py ref + fixture,
CPU hours in the build container).  The same comparison rides in every default bench.py line as `parity_full`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_whole_config2_batch_against_the_reference(engine):
    import full_parity_c2
    if not os.path.exists(full_parity_c2.FIXTURE):
        pytest.skip("tests/golden/c2_full_ref.npz is not in this tree")
    n = len(np.load(full_parity_c2.FIXTURE)["editDistance"])
    target, reads = full_parity_c2.workload(n)
    b = engine.SharedBatch(reads, target, mode="HW", task="distance", k=-1, device=0)
    try:
        b.run()
        flat = b.results_flat()
        got = full_parity_c2.compare_with_fixture(flat, target, reads)
    finally:
        b.close()
    assert got is not None and got["checked"] == n, got
    assert got["bit_exact"] == n, got
