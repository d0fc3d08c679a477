"""Materialise the inputs of a golden-fixture case (shared by oracle/gen_golden.py,
which records the compiled reference's answers, and by the tests, which replay
them on the GPU box where /root/reference does not exist).

A case is a dict with an "input" recipe:
  {"kind": "hex",  "q": "<hex>", "t": "<hex>"}
  {"kind": "rand", "seed": s, "m": m, "tn": tn, "sigma": g}            unrelated sequences
  {"kind": "mut",  "seed": s, "tn": tn, "sigma": 4, "sub": .., "ins": .., "del": ..}
        query = mutated copy of a uniform DNA target (NW shapes, configs 4/5)
  {"kind": "read", "seed": s, "tn": tn, "m": m, "index": i, "tseed": ts}
        read i of synth.illumina_reads over synth.random_dna(ts, tn) (config 2 shape)
plus "mode", "task", "k", "eq" (list of [a, b] single-char strings, latin-1).
"""
import functools

import numpy as np

from edlib_amd import synth


@functools.lru_cache(maxsize=4)
def _target(tseed, tn):
    return synth.random_dna(tseed, tn)


@functools.lru_cache(maxsize=4)
def _reads(tseed, tn, seed, m, n):
    return synth.illumina_reads(_target(tseed, tn), n, m=m, seed=seed)


def materialise(case):
    inp = case["input"]
    kind = inp["kind"]
    if kind == "hex":
        return bytes.fromhex(inp["q"]), bytes.fromhex(inp["t"])
    if kind == "rand":
        q = synth.random_symbols(inp["seed"], inp["m"], inp["sigma"], stream=1)
        t = synth.random_symbols(inp["seed"], inp["tn"], inp["sigma"], stream=2)
        return q.tobytes(), t.tobytes()
    if kind == "mut":
        t = synth.random_dna(inp["seed"], inp["tn"], stream=7)
        q, _ = synth.mutate(t, inp["seed"], inp["sub"], inp["ins"], inp["del"], stream=8)
        if len(q) == 0:
            q = t[:1]
        return q.tobytes(), t.tobytes()
    if kind == "read":
        n = inp.get("n", 64)
        r = _reads(inp["tseed"], inp["tn"], inp["seed"], inp["m"], n)
        return r["reads"][inp["index"]].tobytes(), _target(inp["tseed"], inp["tn"]).tobytes()
    raise ValueError(kind)


def eq_pairs(case):
    eq = case.get("eq")
    if not eq:
        return None
    return [(a.encode("latin-1"), b.encode("latin-1")) for a, b in eq]


def expected(case):
    """The reference's answer with `alignment` as bytes (stored run-length encoded)."""
    ref = dict(case["ref"])
    if ref.get("alignment_rle") is not None:
        out = bytearray()
        for op, run in ref["alignment_rle"]:
            out += bytes([op]) * run
        ref["alignment"] = bytes(out)
    else:
        ref["alignment"] = None
    ref.pop("alignment_rle", None)
    return ref


def rle(ops):
    if ops is None:
        return None
    out = []
    for op in ops:
        if out and out[-1][0] == op:
            out[-1][1] += 1
        else:
            out.append([op, 1])
    return out
