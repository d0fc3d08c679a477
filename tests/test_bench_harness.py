"""bench.py as a harness: the --gpus contract, the multi-rank path with the ENGINE on every rank, and the
configs 4 / 5 lines with their full-batch parity.  CPU part: argument plumbing and the checker's arithmetic
(the reference pool posing as the engine).  GPU part (-m gpu): real runs, two ranks sharing the one device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=1200):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _line(p):
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_gpus_flag_must_match_the_world_size():
    """`--gpus 3` inside a 2-rank launch would report a wrong n_gpus: refused before anything runs"""
    p = _run(["--gpus", "3"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


def test_checker_arithmetic(ref):
    """the parity / invariant code of bench.py on results that are right by construction (the reference pool),
    then with one distance and one end location falsified"""
    if ref is None:
        pytest.skip("oracle/_ref did not travel")
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    old = bench.TARGET_LEN
    bench.TARGET_LEN = 60000
    try:
        w = bench.make_workload(2, 1500, 0, 1, False)
        r = O.pool_align(w["qpool"], w["qoff"], w["tpool"], w["toff"], True, "HW", "distance")
        flat = {k: r[k] for k in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "alnOff", "alignment")}
        flat["starts"] = None
        inv = bench.invariants_config2(w, flat)
        assert inv["ed_le_planted_edits"] and inv["ed_le_read_len"] and inv["planted_position_in_end_locations"]
        assert inv["planted_position_reads_checked"] > 1000
        base, par = bench.cpu_baseline_and_parity(w, flat, 1500)
        assert par["checked"] == 1500 and par["bit_exact"] == 1500 and base["kind"] == "reference" and base["value"] > 0
        flat["ends"] = flat["ends"].copy(); flat["ends"][7] += 1
        base, par = bench.cpu_baseline_and_parity(w, flat, 1500)
        assert par["bit_exact"] == 1499
    finally:
        bench.TARGET_LEN = old
    w = bench.make_workload(5, 64, 0, 1, False)
    r = O.pool_align(w["qpool"], w["qoff"], w["tpool"], w["toff"], False, "NW", "path", want_cigar=True)
    flat = {k: r[k] for k in ("status", "editDistance", "numLocations", "alphabetLength", "locOff", "ends", "starts", "alnOff", "alignment")}
    base, par = bench.cpu_baseline_and_parity(w, flat, 64)
    assert par["bit_exact"] == 64 and par["cigar_extended_equal"] and par["cigar_standard_equal"] and par["op_bytes_equal"]


# ------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_two_engine_ranks_agree_with_one(tmp_path):
    """bench.py --gpus 2 spawns two ranks itself (torch.distributed.run, gloo, both on device 0 of this 1-GPU box);
    each runs the HIP engine on its shard; the gathered distances equal a one-rank run of the same global batch"""
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    common = ["--reads", "6000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--strong"]
    a = _line(_run(common + ["--dump", one]))
    b = _line(_run(common + ["--gpus", "2", "--share-gpu", "--dump", two]))
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["dry_run_shared_gpu"] and len(b["per_rank_ms_per_step"]) == 2
    assert b["scaling"] == "strong" and b["devices_distinct"] == 1
    x, y = np.load(one), np.load(two)
    assert len(x) == 6000 and np.array_equal(x, y)
    # weak scaling: rank r generates its own batch (seed + r); rank 0's half equals the one-rank run of 3000 reads
    c = _line(_run(["--reads", "3000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--gpus", "2",
                    "--share-gpu", "--dump", two]))
    d = _line(_run(["--reads", "3000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--dump", one]))
    assert c["scaling"] == "weak" and c["n_gpus"] == 2 and d["n_gpus"] == 1
    assert np.array_equal(np.load(two)[:3000], np.load(one))


@pytest.mark.gpu
def test_gpus_2_without_a_second_device_fails_loudly():
    import edlib_amd
    if edlib_amd.device_count() >= 2:
        pytest.skip("this box has a second GPU")
    p = _run(["--gpus", "2", "--reads", "2000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"])
    assert p.returncode != 0 and "only 1 device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,units", [(2, 4096), (4, 600), (5, 1500)])
def test_config_lines_carry_full_parity(cfg, units):
    out = _line(_run(["--config", str(cfg), "--units", str(units), "--steps", "1", "--warmup", "1", "--parity-sample", str(units)]))
    assert out["config"]["baseline_config"] == cfg and out["n_gpus"] == 1
    for key in ("roofline", "valu_roofline", "cpu_baseline", "parity_sample"):
        assert key in out
    assert out["parity_sample"]["checked"] == units and out["parity_sample"]["bit_exact"] == units
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("reference", "port")
    if cfg == 5:
        assert out["parity_sample"]["cigar_extended_equal"] and out["parity_sample"]["cigar_standard_equal"]
    if cfg == 2:
        assert all(v is True for k, v in out["invariants"].items() if k != "planted_position_reads_checked")
        assert out["e2e"]["distances_equal_resident"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("strong", [False, True])
def test_eight_ranks_dry_run_on_one_device(strong):
    """the driver's 8-GPU launch shape on a 1-GPU box: eight ranks (torch.distributed.run, gloo, all on device 0), weak
    and strong scaling; every rank reports, host pools are sized from the CPU quota divided by the world size, and the
    run finishes well inside two minutes"""
    import time
    args = ["--reads", "64000" if strong else "8000", "--steps", "2", "--warmup", "1", "--parity-sample", "400", "--no-e2e",
            "--gpus", "8", "--share-gpu"] + (["--strong"] if strong else [])
    t0 = time.time()
    out = _line(_run(args, timeout=300))
    wall = time.time() - t0
    assert out["n_gpus"] == 8 and out["dry_run_shared_gpu"] and len(out["per_rank_ms_per_step"]) == 8
    assert out["scaling"] == ("strong" if strong else "weak") and out["devices_distinct"] == 1
    assert out["config"]["units_per_gpu"] == 8000
    # a multi-rank line carries its evidence: the reference over a sample of rank 0's shard, after the timed region
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] in ("reference", "port")
    assert out["parity_sample"]["checked"] >= 400 and out["parity_sample"]["bit_exact"] == out["parity_sample"]["checked"]
    assert out["parity_sample"]["shard"] == "rank 0 of 8" and all(v is True for k, v in out["invariants"].items() if k != "planted_position_reads_checked")
    assert wall < 150, wall
