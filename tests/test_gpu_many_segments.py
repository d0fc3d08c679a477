"""-m gpu: a small batch of reads against a long target is cut into hundreds of target segments (every SIMD wants a wave);
the per-segment results of a read are merged by a wave per read (reads_kernels.hip: merge_segments_wave_kernel, S >= 64).
Repeats make the merge matter: end locations of one read in many segments, in target order, lists that fit the 16 kept per
read and lists that do not (the exact second pass, itself on short segments when few reads need it)."""
import numpy as np
import pytest

from edlib_amd import synth
from test_gpu_long_reads import _check, _reads

pytestmark = pytest.mark.gpu


def _target_with_repeats(seed, n, copies):
    rng = np.random.default_rng(seed)
    t = synth.random_dna(seed, n).copy()
    motifs = []
    for j, c in enumerate(copies):
        motif = synth.random_dna(seed + 1 + j, 400)
        for at in rng.integers(0, n - 400, c):
            t[at:at + 400] = motif
        motifs.append(motif)
    return t, motifs


@pytest.mark.parametrize("task", ["distance", "locations", "path"])
def test_reads_with_end_locations_in_many_segments(engine, task):
    target, motifs = _target_with_repeats(301, 1_200_000, [40, 6, 2])
    rng = np.random.default_rng(302)
    n = 420 if task == "distance" else 180
    reads = _reads(target, [int(x) for x in rng.integers(60, 160, n)], 303, unrelated_every=11, max_err=0.05)
    for j, motif in enumerate(motifs):                      # reads out of the repeats: 40 (overflow), 6 and 2 end locations
        for s in (0, 37, 120, 250):
            m = int(rng.integers(80, 150))
            reads.append(np.ascontiguousarray(motif[s:s + m]))
            r = motif[s:s + m].copy(); r[m // 2] = ord("A") if r[m // 2] != ord("A") else ord("C")
            reads.append(r)
    st = _check(engine, reads, target, task)
    assert st["path"] & 1
    assert st["overflow_units"] >= 4, st                    # the 40-copy repeat went through the exact pass
