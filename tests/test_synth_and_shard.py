"""CPU: the synthetic generators are deterministic (fixtures depend on it) and the
multi-GPU sharding helpers work over gloo with world_size 2."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from edlib_amd import synth
from edlib_amd.parallel import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_pinned():
    t = synth.random_dna(12345, 100000)
    assert bytes(t[:40]) == b"AGGTGCCCCCGACGACACGAGAGGATGTGCCTCCGGTCTA"
    r = synth.illumina_reads(t, 3000, m=150, seed=12346)
    assert hashlib.sha256(r["reads"].tobytes()).hexdigest()[:16] == PINNED_READS
    r2 = synth.illumina_reads(t, 1000, m=150, seed=12346, chunk=77)
    assert (r2["reads"] == r["reads"][:1000]).all() and (r2["edits"] == r["edits"][:1000]).all()
    q, n = synth.mutate(t[:1000], 5, 0.03, 0.01, 0.01)
    assert abs(len(q) - 1000) < 40 and 20 < n < 90


PINNED_READS = "48c744e174106f26"


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            got = [shard_range(n, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from edlib_amd.parallel import aggregate_throughput, gather_int_results, shard_range
    from oracle.oracle import load_oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = load_oracle()                       # stands in for the GPU engine in this CPU test
    target = synth.random_dna(3, 3000)
    reads = synth.illumina_reads(target, 37, m=60, seed=4)["reads"]
    lo, hi = shard_range(len(reads), rank, world)
    local = [orc.align(reads[i].tobytes(), target.tobytes(), "HW", "distance", -1)["editDistance"] for i in range(lo, hi)]
    full = gather_int_results(local, len(reads), dist)
    cells, secs = aggregate_throughput((hi - lo) * 60 * 3000, 1.0 + rank, dist)
    if rank == 0:
        np.save(os.path.join(tmp, "full.npy"), full)
        np.save(os.path.join(tmp, "agg.npy"), np.array([cells, secs]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path, oracle):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    full = np.load(tmp_path / "full.npy")
    target = synth.random_dna(3, 3000)
    reads = synth.illumina_reads(target, 37, m=60, seed=4)["reads"]
    want = [oracle.align(reads[i].tobytes(), target.tobytes(), "HW", "distance", -1)["editDistance"] for i in range(37)]
    assert full.tolist() == want
    cells, secs = np.load(tmp_path / "agg.npy")
    assert cells == 37 * 60 * 3000 and secs == 2.0


def test_strong_scaling_shards_partition_the_global_batch():
    """bench.py --strong: every rank generates the same global batch and keeps its contiguous slice;
    the slices must tile the batch exactly (no read twice, none dropped), for sizes that do not divide."""
    from edlib_amd.parallel import shard_range
    target = synth.random_dna(12345, 4000)
    n = 1003
    full = synth.illumina_reads(target, n, m=50, seed=12346)
    for world in (1, 2, 3, 8):
        parts, covered = [], 0
        for rank in range(world):
            lo, hi = shard_range(n, rank, world)
            assert lo == covered and hi >= lo
            covered = hi
            again = synth.illumina_reads(target, n, m=50, seed=12346)      # what that rank would generate
            parts.append(again["reads"][lo:hi])
        assert covered == n
        assert np.array_equal(np.concatenate(parts), full["reads"])


def test_sources_digest_follows_the_product_sources(tmp_path):
    """profiles/hbm_traffic.json carries the digest of the sources its counters were taken on; bench.py compares it with the
    running tree's (`roofline.traffic_source.sources_identical`)"""
    from edlib_amd.parallel import sources_sha
    (tmp_path / "edlib_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (tmp_path / "bench.py").write_text("x = 1\n")
    (tmp_path / "edlib_amd" / "csrc" / "k.hip").write_text("kernel\n")
    (tmp_path / "README.md").write_text("docs\n")
    a = sources_sha(str(tmp_path))
    (tmp_path / "README.md").write_text("other docs\n")
    assert sources_sha(str(tmp_path)) == a                       # documents do not count
    (tmp_path / "edlib_amd" / "csrc" / "k.hip").write_text("kernel 2\n")
    assert sources_sha(str(tmp_path)) != a
