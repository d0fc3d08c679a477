#!/usr/bin/env python3
"""Whole-batch parity of BASELINE config 2 ("weak #1": the headline was pinned on a 2 % sample).

  python tools/full_parity_c2.py ref  [--threads 6] [--out gpurun_out/c2ref] [--units 1000000]
        CPU, hours: the UNMODIFIED reference (oracle/_ref/libedlib_ref.so through oracle/ref_pool.cpp) over
        ALL reads of bench.py's config-2 batch (rank 0, weak scaling: seed 12346), in resumable chunks.
  python tools/full_parity_c2.py gpu  [--out gpurun_out/c2gpu.npz] [--units 1000000]
        GPU box: the engine's results for the same batch (editDistance, numLocations, alphabetLength, every end
        location), written as one compressed file that travels back through gpurun_out/.
  python tools/full_parity_c2.py compare --ref gpurun_out/c2ref --gpu gpurun_out/c2gpu.npz --json profiles/r03_c2_full_parity.json
        field-by-field comparison; failing reads are listed (they become fixtures).

Test infrastructure: the only user of oracle/ here is the `ref` leg (the checker), never the product path.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TARGET_LEN, READ_LEN = 5_000_000, 150


def workload(units):
    from edlib_amd import synth
    target = synth.random_dna(12345, TARGET_LEN)
    rd = synth.illumina_reads(target, units, m=READ_LEN, seed=12346)
    return target, np.ascontiguousarray(rd["reads"])


def digest(target, reads):
    h = hashlib.sha256(); h.update(target.tobytes()); h.update(reads.tobytes())
    return h.hexdigest()


def leg_ref(args):
    from oracle import oracle as O
    lib, kind = O.checker_library()
    assert kind == "reference", "the compiled reference (oracle/_ref) is needed for this leg"
    target, reads = workload(args.units)
    n = len(reads)
    os.makedirs(args.out, exist_ok=True)
    meta = {"units": n, "sha256": digest(target, reads), "chunk": args.chunk}
    json.dump(meta, open(os.path.join(args.out, "meta.json"), "w"))
    qoff = np.arange(n + 1, dtype=np.int64) * READ_LEN
    toff = np.array([0, TARGET_LEN], dtype=np.int64)
    t0 = time.time()
    for a in range(0, n, args.chunk):
        path = os.path.join(args.out, "chunk_%07d.npz" % a)
        if os.path.exists(path):
            continue
        b = min(n, a + args.chunk)
        sel = np.arange(a, b, dtype=np.int32)
        r = O.pool_align(reads.reshape(-1), qoff, target, toff, True, "HW", "distance", -1, select=sel,
                         threads=args.threads, libpath=lib)
        np.savez_compressed(path + ".tmp.npz", editDistance=r["editDistance"], numLocations=r["numLocations"],
                            alphabetLength=r["alphabetLength"], status=r["status"], locOff=r["locOff"], ends=r["ends"],
                            wall=r["wall_seconds"])
        os.replace(path + ".tmp.npz", path)
        print("[ref] %d..%d done, %.0f s elapsed" % (a, b, time.time() - t0), flush=True)


def leg_gpu(args):
    import edlib_amd
    target, reads = workload(args.units)
    b = edlib_amd.SharedBatch(reads, target, mode="HW", task="distance", k=-1, device=0)
    b.run()
    f = b.results_flat()
    b.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, editDistance=f["editDistance"].astype(np.int32), numLocations=f["numLocations"].astype(np.int32),
                        alphabetLength=f["alphabetLength"].astype(np.int32), status=f["status"].astype(np.int32),
                        locOff=f["locOff"].astype(np.int64), ends=f["ends"].astype(np.int32),
                        sha256=np.frombuffer(digest(target, reads).encode(), dtype=np.uint8),
                        commit=np.frombuffer((open(os.path.join(ROOT, ".visit_commit")).read().strip() if os.path.exists(os.path.join(ROOT, ".visit_commit")) else "unknown").encode(), dtype=np.uint8))
    print("[gpu] %d units written to %s" % (len(reads), args.out))


def leg_compare(args):
    meta = json.load(open(os.path.join(args.ref, "meta.json")))
    g = np.load(args.gpu)
    gsha = bytes(g["sha256"]).decode()
    n = meta["units"]
    out = {"config": 2, "units": n, "engine_commit": bytes(g["commit"]).decode() if "commit" in g.files else "7486eea (first visit of round 3)",
           "inputs_sha256": meta["sha256"], "inputs_equal": gsha == meta["sha256"],
           "fields": "status, editDistance, numLocations, alphabetLength, every endLocation", "checked": 0, "bit_exact": 0,
           "failing_units": [], "reference": "oracle/_ref/libedlib_ref.so (unmodified /root/reference/edlib/src/edlib.cpp)"}
    wall = 0.0
    gl = g["locOff"]
    for a in range(0, n, meta["chunk"]):
        path = os.path.join(args.ref, "chunk_%07d.npz" % a)
        if not os.path.exists(path):
            continue
        r = np.load(path)
        b = a + len(r["editDistance"])
        bad = np.zeros(b - a, dtype=bool)
        for f in ("status", "editDistance", "numLocations", "alphabetLength"):
            bad |= g[f][a:b] != r[f]
        cnt_g = gl[a + 1:b + 1] - gl[a:b]
        cnt_r = r["locOff"][1:] - r["locOff"][:-1]
        bad |= cnt_g != cnt_r
        ok = ~bad
        if ok.all():
            same = g["ends"][gl[a]:gl[b]] == r["ends"]
            first = np.cumsum(cnt_r) - cnt_r
            cs = np.concatenate([[0], np.cumsum(~same)])
            bad |= (cs[first + cnt_r] - cs[first]) > 0
        else:                                     # slow path: unit by unit
            for i in np.nonzero(ok)[0]:
                if not np.array_equal(g["ends"][gl[a + i]:gl[a + i + 1]], r["ends"][r["locOff"][i]:r["locOff"][i + 1]]):
                    bad[i] = True
        out["checked"] += b - a
        out["bit_exact"] += int((~bad).sum())
        out["failing_units"] += [int(a + i) for i in np.nonzero(bad)[0][:50]]
        wall += float(r["wall"])
    out["reference_wall_seconds"] = round(wall, 1)
    print(json.dumps(out))
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


FIXTURE = os.path.join(ROOT, "tests", "golden", "c2_full_ref.npz")


def leg_fixture(args):
    """the reference's answers for the WHOLE batch as one compact committed fixture: what bench.py's
    `parity_full` and tests/test_gpu_full_parity_c2.py compare the resident results of the timed batch with"""
    meta = json.load(open(os.path.join(args.ref, "meta.json")))
    n = meta["units"]
    ed, nl, al, ends = [], [], [], []
    for a in range(0, n, meta["chunk"]):
        r = np.load(os.path.join(args.ref, "chunk_%07d.npz" % a))              # (a missing chunk is an error: the fixture is whole)
        assert (r["status"] == 0).all()
        ed.append(r["editDistance"].astype(np.int16)); nl.append(r["numLocations"].astype(np.int32))
        al.append(r["alphabetLength"].astype(np.uint8)); ends.append(r["ends"].astype(np.int32))
    out = args.out if args.out and args.out.endswith(".npz") else FIXTURE
    np.savez_compressed(out, editDistance=np.concatenate(ed), numLocations=np.concatenate(nl), alphabetLength=np.concatenate(al),
                        ends=np.concatenate(ends), sha256=np.frombuffer(meta["sha256"].encode(), dtype=np.uint8),
                        note=np.frombuffer(b"edlibAlign(HW, DISTANCE, k = -1) of the UNMODIFIED reference (oracle/_ref) over bench.py's config-2 "
                                           b"batch, rank 0, weak scaling (target seed 12345, reads seed 12346); made by tools/full_parity_c2.py ref + fixture", dtype=np.uint8))
    print("[fixture] %d units -> %s (%.1f MB)" % (n, out, os.path.getsize(out) / 1e6))


def compare_with_fixture(flat, target, reads, fixture=FIXTURE):
    """whole-batch comparison of results_flat() arrays with the committed reference answers; None when the fixture is absent
    or was made for other inputs"""
    if not os.path.exists(fixture):
        return None
    f = np.load(fixture)
    n = len(f["editDistance"])
    if len(reads) != n or bytes(f["sha256"]).decode() != digest(target, np.ascontiguousarray(reads)):
        return {"checked": 0, "note": "fixture is for other inputs"}
    bad = (flat["editDistance"].astype(np.int64) != f["editDistance"]) | (flat["numLocations"].astype(np.int64) != f["numLocations"]) | \
          (flat["alphabetLength"].astype(np.int64) != f["alphabetLength"]) | (flat["status"].astype(np.int64) != 0)
    off = flat["locOff"].astype(np.int64)
    roff = np.concatenate([[0], np.cumsum(f["numLocations"].astype(np.int64))])
    if not bad.any() and off[-1] == roff[-1]:
        same = flat["ends"].astype(np.int64)[: off[-1]] == f["ends"]
        cs = np.concatenate([[0], np.cumsum(~same)])
        bad |= (cs[roff[1:]] - cs[roff[:-1]]) > 0
    else:
        ge, fe = flat["ends"], f["ends"]
        for i in np.nonzero(~bad)[0]:
            if not np.array_equal(ge[off[i]:off[i + 1]], fe[roff[i]:roff[i + 1]]):
                bad[i] = True
    return {"checked": int(n), "bit_exact": int((~bad).sum()), "failing_units": [int(i) for i in np.nonzero(bad)[0][:20]],
            "fields": "status, editDistance, numLocations, alphabetLength, every endLocation",
            "reference": "tests/golden/c2_full_ref.npz: the unmodified reference over the whole batch (tools/full_parity_c2.py ref, CPU hours)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("leg", choices=["ref", "gpu", "compare", "fixture"])
    ap.add_argument("--units", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=20000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ref", default=os.path.join(ROOT, "gpurun_out", "c2ref"))
    ap.add_argument("--gpu", default=os.path.join(ROOT, "gpurun_out", "c2gpu.npz"))
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    if args.out is None and args.leg != "fixture":
        args.out = args.ref if args.leg == "ref" else args.gpu
    {"ref": leg_ref, "gpu": leg_gpu, "compare": leg_compare, "fixture": leg_fixture}[args.leg](args)


if __name__ == "__main__":
    main()
