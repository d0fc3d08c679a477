"""edlib_amd -- Python front of the MI355X edit-distance engine (ctypes over the C ABI).

Mirrors the reference's Python binding (bindings/python/edlib.pyx): ``align()`` and
``getNiceAlignment()`` have the same arguments, defaults, result dictionary and
error behaviour (edlib.pyx:56-155, 158-238), so the reference's own binding tests
(bindings/python/test.py) read the same against this package.  Additive:
``align_batch()`` / ``align_pairs()`` and the resident ``SharedBatch`` / ``PairBatch``
sessions over include/edlib_amd.h.

There is no CPU path in here: everything calls ``libedlib.so`` (built by
``__graft_entry__.build()`` / ``make``), and a missing library or a missing GPU
raises.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EDLIB_AMD_LIB: another build of the same C ABI (A/B timing of two builds on one box: tools/gpu_visit.sh ab)
LIB_PATH = os.environ.get("EDLIB_AMD_LIB") or os.path.join(_HERE, "libedlib.so")

EDLIB_MODE = {"NW": 0, "SHW": 1, "HW": 2}
EDLIB_TASK = {"distance": 0, "locations": 1, "path": 2}


class EqualityPair(C.Structure):          # edlib.h:92-95
    _fields_ = [("first", C.c_char), ("second", C.c_char)]


class AlignConfig(C.Structure):           # edlib.h:100-140
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int),
                ("additionalEqualities", C.POINTER(EqualityPair)),
                ("additionalEqualitiesLength", C.c_int)]


class AlignResult(C.Structure):           # edlib.h:162-218
    _fields_ = [("status", C.c_int), ("editDistance", C.c_int),
                ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)),
                ("numLocations", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)),
                ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]


class BatchStats(C.Structure):            # edlib_amd.h EdlibAmdBatchStats
    _fields_ = [("run_ms", C.c_double), ("scan_ms", C.c_double), ("scan_launches", C.c_int),
                ("cells", C.c_longlong), ("word_steps", C.c_longlong), ("algo_bytes", C.c_longlong),
                ("path", C.c_int), ("overflow_units", C.c_int), ("wide_retries", C.c_int)]


class ResultsView(C.Structure):          # edlib_amd.h EdlibAmdResultsView
    _fields_ = [("numUnits", C.c_int), ("status", C.POINTER(C.c_int)), ("editDistance", C.POINTER(C.c_int)),
                ("numLocations", C.POINTER(C.c_int)), ("alphabetLength", C.POINTER(C.c_int)),
                ("locOffsets", C.POINTER(C.c_longlong)), ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)), ("alnOffsets", C.POINTER(C.c_longlong)),
                ("alignment", C.POINTER(C.c_ubyte))]


_lib = None


def lib():
    """The loaded C-ABI library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s not found: build it with `make` or __graft_entry__.build() "
                          "(there is no Python/CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.edlibAlign.restype = AlignResult
        L.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, AlignConfig]
        L.edlibNewAlignConfig.restype = AlignConfig
        L.edlibNewAlignConfig.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(EqualityPair), C.c_int]
        L.edlibDefaultAlignConfig.restype = AlignConfig
        L.edlibFreeAlignResult.argtypes = [AlignResult]
        L.edlibAlignmentToCigar.restype = C.c_void_p
        L.edlibAlignmentToCigar.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.edlibAmdDeviceCount.restype = C.c_int
        L.edlibAmdLastError.restype = C.c_char_p
        L.edlibAmdVersion.restype = C.c_char_p
        L.edlibAmdBatchCreateShared.restype = C.c_void_p
        L.edlibAmdBatchCreateShared.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                AlignConfig, C.c_int]
        L.edlibAmdBatchCreatePairs.restype = C.c_void_p
        L.edlibAmdBatchCreatePairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               AlignConfig, C.c_int]
        L.edlibAmdBatchRun.argtypes = [C.c_void_p]
        L.edlibAmdBatchResults.argtypes = [C.c_void_p, C.POINTER(AlignResult)]
        L.edlibAmdBatchResultsFlat.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.edlibAmdBatchResultsView.argtypes = [C.c_void_p, C.POINTER(ResultsView)]
        L.edlibAmdBatchCigarView.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.edlibAmdFreeResults.argtypes = [C.POINTER(AlignResult), C.c_int]
        L.edlibAmdFreeResults.restype = None
        L.edlibAmdTrim.restype = None
        L.edlibAmdBatchStats.argtypes = [C.c_void_p, C.POINTER(BatchStats)]
        L.edlibAmdBatchDestroy.argtypes = [C.c_void_p]
        L.edlibAmdBatchDestroy.restype = None
        L.edlibAlignBatchSharedTarget.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int,
                                                  AlignConfig, C.POINTER(AlignResult)]
        L.edlibAlignBatchPairs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_int), C.c_int, AlignConfig, C.POINTER(AlignResult)]
        L.libc = C.CDLL(None)
        L.libc.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def device_count():
    return lib().edlibAmdDeviceCount()


def last_error():
    return lib().edlibAmdLastError().decode()


# --------------------------------------------------------------- helpers

class NeedsAlphabetMapping(Exception):
    pass


def _map_ascii_string(s):
    if isinstance(s, (bytes, bytearray)):
        return bytes(s)
    if isinstance(s, np.ndarray) and s.dtype == np.uint8:
        return s.tobytes()
    if isinstance(s, str):
        b = s.encode("utf-8")
        if len(b) == len(s):
            return b
    raise NeedsAlphabetMapping()


def _map_to_bytes(query, target, additional_equalities):
    """Arbitrary hashable symbols -> single bytes (edlib.pyx:22-53)."""
    try:
        return _map_ascii_string(query), _map_ascii_string(target), additional_equalities
    except NeedsAlphabetMapping:
        alphabet = set(query).union(set(target))
        if len(alphabet) > 256:
            raise ValueError("query and target combined have more than 256 unique values, "
                             "this is not supported.")
        mapping = {c: bytes([i]) for i, c in enumerate(alphabet)}
        q = b"".join(mapping[c] for c in query)
        t = b"".join(mapping[c] for c in target)
        if additional_equalities is not None:
            additional_equalities = [(mapping[a], mapping[b]) for a, b in additional_equalities
                                     if a in mapping and b in mapping]
        return q, t, additional_equalities


def _one_byte(x):
    if isinstance(x, (bytes, bytearray)):
        return bytes(x[:1])
    return x.encode("utf-8")[:1]


def _make_config(mode, task, k, additionalEqualities):
    cfg = lib().edlibDefaultAlignConfig()
    if k is not None:
        cfg.k = k
    if mode in EDLIB_MODE:
        cfg.mode = EDLIB_MODE[mode]
    elif isinstance(mode, int):
        cfg.mode = mode
    if task in EDLIB_TASK:
        cfg.task = EDLIB_TASK[task]
    elif isinstance(task, int):
        cfg.task = task
    keep = None
    if additionalEqualities:
        keep = (EqualityPair * len(additionalEqualities))()
        for i, (a, b) in enumerate(additionalEqualities):
            keep[i].first = _one_byte(a)
            keep[i].second = _one_byte(b)
        cfg.additionalEqualities = C.cast(keep, C.POINTER(EqualityPair))
        cfg.additionalEqualitiesLength = len(additionalEqualities)
    return cfg, keep


def cigar_from_alignment(ops, extended=True):
    """edlibAlignmentToCigar (edlib.h:268-271) on a bytes object of op codes."""
    L = lib()
    p = L.edlibAlignmentToCigar(bytes(ops), len(ops), 1 if extended else 0)
    if not p:
        return None
    s = C.string_at(p).decode()
    L.libc.free(p)
    return s


def _raw_result(r):
    """EdlibAlignResult -> dict with every C field (used by the parity tests)."""
    n = r.numLocations
    return {
        "status": r.status,
        "editDistance": r.editDistance,
        "endLocations": [r.endLocations[i] for i in range(n)] if r.endLocations else None,
        "startLocations": [r.startLocations[i] for i in range(n)] if r.startLocations else None,
        "numLocations": n,
        "alignment": bytes(bytearray(r.alignment[:r.alignmentLength])) if r.alignment else None,
        "alignmentLength": r.alignmentLength,
        "alphabetLength": r.alphabetLength,
    }


def _nice_result(raw):
    """The reference binding's dictionary (edlib.pyx:136-153)."""
    locations = []
    for i in range(max(raw["numLocations"], 0)):
        locations.append((raw["startLocations"][i] if raw["startLocations"] is not None else None,
                          raw["endLocations"][i] if raw["endLocations"] is not None else None))
    cigar = cigar_from_alignment(raw["alignment"]) if raw["alignment"] is not None else None
    return {"editDistance": raw["editDistance"], "alphabetLength": raw["alphabetLength"],
            "locations": locations, "cigar": cigar}


# ----------------------------------------------------------- public API

def align_raw(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None):
    """edlibAlign() with every field of EdlibAlignResult returned (bytes in)."""
    L = lib()
    cfg, keep = _make_config(mode, task, k, additionalEqualities)
    r = L.edlibAlign(query, len(query), target, len(target), cfg)
    raw = _raw_result(r)
    L.edlibFreeAlignResult(r)
    return raw


def align(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None):
    """Align query with target using edit distance (same contract as the reference's
    ``edlib.align``, edlib.pyx:56-155).  Returns {editDistance, alphabetLength,
    locations: [(start, end)], cigar}; raises on status == 1."""
    q, t, eqs = _map_to_bytes(query, target, additionalEqualities)
    raw = align_raw(q, t, mode, task, k, eqs)
    if raw["status"] == 1:
        raise Exception("There was an error.")
    return _nice_result(raw)


# how each extended-CIGAR op fills the three display rows: (takes a query symbol, takes a target symbol, marker)
_NICE_OPS = {"=": (True, True, "|"), "X": (True, True, "."), "I": (True, False, None), "D": (False, True, None)}


def getNiceAlignment(alignResult, query, target, gapSymbol="-"):
    """Three display rows for a result of ``align(..., task="path")``: the query and target with gap symbols
    inserted, and between them a row of '|' (match), '.' (mismatch) and gap symbols.

    Same call and result keys as the reference binding's helper (bindings/python/edlib.pyx:158-238:
    ``query_aligned`` / ``matched_aligned`` / ``target_aligned``); like it, raises ``Exception`` when the
    argument is not an ``align()`` dictionary with a CIGAR.  The rows are assembled column by column from
    the expanded CIGAR; the target row starts at the first reported start location (0 when there is none)."""
    if not isinstance(alignResult, dict) or type(alignResult) is not dict:
        raise Exception("getNiceAlignment() needs the dictionary returned by align().")
    for key in ("locations", "cigar"):
        if key not in alignResult:
            raise Exception("getNiceAlignment(): the align() result has no '%s' entry." % key)
    cigar = alignResult["cigar"]
    if not cigar:
        raise Exception("getNiceAlignment(): empty CIGAR -- run align() with task='path'.")
    runs = re.findall(r"(\d+)(\D)", cigar)
    if any(op not in _NICE_OPS for _, op in runs) or "".join(n + op for n, op in runs) != cigar:
        raise Exception("getNiceAlignment(): the CIGAR must be in the extended format (=, X, I, D only).")
    start = alignResult["locations"][0][0] if alignResult["locations"] else None
    qi, ti = 0, (start or 0)
    rows = ([], [], [])                       # query, markers, target
    for count, op in runs:
        count = int(count)
        from_q, from_t, mark = _NICE_OPS[op]
        gaps = gapSymbol * count
        rows[0].append(query[qi:qi + count] if from_q else gaps)
        rows[2].append(target[ti:ti + count] if from_t else gaps)
        rows[1].append(mark * count if mark else gaps)
        qi += count if from_q else 0
        ti += count if from_t else 0
    return {"query_aligned": "".join(rows[0]), "matched_aligned": "".join(rows[1]), "target_aligned": "".join(rows[2])}


# ------------------------------------------------------------ batches

def _pack(seqs):
    """list of bytes / uint8 arrays (or one 2-D uint8 array) -> (contiguous uint8, int64 offsets)."""
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2 and seqs.dtype == np.uint8:
        n, m = seqs.shape
        return np.ascontiguousarray(seqs).reshape(-1), np.arange(n + 1, dtype=np.int64) * m
    arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.asarray(s, dtype=np.uint8)
            for s in seqs]
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs])
    data = np.concatenate(arrs) if arrs and off[-1] > 0 else np.zeros(1, dtype=np.uint8)
    return np.ascontiguousarray(data), off


class _Batch:
    """A batch resident in HBM: create (upload) once, run() many times, results()."""

    def __init__(self, handle, n, keep):
        if not handle:
            raise RuntimeError("edlib_amd: batch creation failed: " + last_error())
        self._h = handle
        self.n = n
        self._keep = keep

    def run(self):
        if lib().edlibAmdBatchRun(self._h) != 0:
            raise RuntimeError("edlib_amd: run failed: " + last_error())
        return self.stats()

    def stats(self):
        s = BatchStats()
        lib().edlibAmdBatchStats(self._h, C.byref(s))
        return {f: getattr(s, f) for f, _ in BatchStats._fields_}

    def results(self, raw=True):
        L = lib()
        arr = (AlignResult * max(self.n, 1))()
        if L.edlibAmdBatchResults(self._h, arr) != 0:
            raise RuntimeError("edlib_amd: results failed: " + last_error())
        out = []
        for i in range(self.n):
            d = _raw_result(arr[i])
            L.edlibFreeAlignResult(arr[i])
            out.append(d if raw else _nice_result(d))
        return out

    def results_flat(self, copy=True):
        """Every field of every result as flat numpy arrays (edlibAmdBatchResultsView: no per-unit malloc):
        status / editDistance / numLocations / alphabetLength [n], locOff [n+1] into ends / starts
        (starts is None unless the task produced start locations), alnOff [n+1] into alignment (op bytes, None unless the
        task produced paths).  copy=False: views of the batch's own pinned memory, valid until its next run() / close()."""
        L = lib()
        n = self.n
        v = ResultsView()
        if L.edlibAmdBatchResultsView(self._h, C.byref(v)) != 0:
            raise RuntimeError("edlib_amd: results failed: " + last_error())

        def arr(ptr, count, dtype):
            if not ptr:
                return None
            if count == 0:
                return np.zeros(0, dtype=dtype)
            a = np.ctypeslib.as_array(ptr, shape=(count,))
            return a.copy() if copy else a
        loc = arr(v.locOffsets, n + 1, np.int64)
        aln = arr(v.alnOffsets, n + 1, np.int64)
        nloc, naln = int(loc[-1]), int(aln[-1])
        ends = arr(v.endLocations, nloc, np.int32)
        starts = arr(v.startLocations, nloc, np.int32)
        ops = arr(v.alignment, naln, np.uint8)
        if ops is None and naln == 0:
            ops = np.zeros(0, dtype=np.uint8)
        return {"status": arr(v.status, n, np.int32), "editDistance": arr(v.editDistance, n, np.int32),
                "numLocations": arr(v.numLocations, n, np.int32), "alphabetLength": arr(v.alphabetLength, n, np.int32),
                "locOff": loc, "ends": ends if ends is not None else np.zeros(0, dtype=np.int32), "starts": starts,
                "alnOff": aln, "alignment": ops}

    def cigars(self, extended=True, copy=True):
        """edlibAlignmentToCigar over every op string of the last run (edlibAmdBatchCigarView; made on the device for a
        batch of short pairs): (chars, off) -- the NUL-terminated strings one after the other as a uint8 array, and the
        int64 offset of every string (off[n] = all bytes)."""
        L = lib()
        pc, po = C.c_void_p(), C.c_void_p()
        if L.edlibAmdBatchCigarView(self._h, 1 if extended else 0, C.byref(pc), C.byref(po)) != 0:
            raise RuntimeError("edlib_amd: cigars failed: " + last_error())
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_longlong)), shape=(self.n + 1,))
        total = int(off[-1])
        chars = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_ubyte)), shape=(max(total, 1),))[:total]
        return (chars.copy(), off.copy()) if copy else (chars, off)

    def cigar_list(self, extended=True):
        chars, off = self.cigars(extended, copy=False)
        raw = chars.tobytes()
        return [raw[int(off[i]):int(off[i + 1]) - 1].decode() for i in range(self.n)]

    def results_arrays(self):
        """editDistance / numLocations / first end location / per-unit end lists (kept for older callers;
        built from results_flat())."""
        f = self.results_flat()
        loc = f["locOff"]
        first = np.full(self.n, -2, dtype=np.int64)
        has = f["numLocations"] > 0
        first[has] = f["ends"][loc[:-1][has]]
        return {"editDistance": f["editDistance"], "numLocations": f["numLocations"], "firstEnd": first,
                "alphabetLength": f["alphabetLength"], "ends": np.split(f["ends"], loc[1:-1])}

    def close(self):
        if self._h:
            lib().edlibAmdBatchDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SharedBatch(_Batch):
    """Many queries against one target (the loop of apps/aligner/aligner.cpp:162-225)."""

    def __init__(self, queries, target, mode="HW", task="distance", k=-1, additionalEqualities=None, device=0):
        qd, qo = _pack(queries)
        t = np.frombuffer(target, dtype=np.uint8) if isinstance(target, (bytes, bytearray)) else np.asarray(target, dtype=np.uint8)
        t = np.ascontiguousarray(t) if len(t) else np.zeros(1, dtype=np.uint8)
        tlen = len(target)
        cfg, keep = _make_config(mode, task, k, additionalEqualities)
        h = lib().edlibAmdBatchCreateShared(qd.ctypes.data, qo.ctypes.data, len(qo) - 1,
                                            t.ctypes.data, tlen, cfg, device)
        super().__init__(h, len(qo) - 1, keep)


class PairBatch(_Batch):
    """Independent (query, target) pairs."""

    def __init__(self, queries, targets, mode="NW", task="distance", k=-1, additionalEqualities=None, device=0):
        qd, qo = _pack(queries)
        td, to = _pack(targets)
        if len(qo) != len(to):
            raise ValueError("queries and targets differ in count")
        cfg, keep = _make_config(mode, task, k, additionalEqualities)
        h = lib().edlibAmdBatchCreatePairs(qd.ctypes.data, qo.ctypes.data, td.ctypes.data, to.ctypes.data,
                                           len(qo) - 1, cfg, device)
        super().__init__(h, len(qo) - 1, keep)


def align_batch(queries, target, mode="HW", task="distance", k=-1, additionalEqualities=None, raw=False):
    """[align(q, target, ...) for q in queries] in one device batch."""
    b = SharedBatch(queries, target, mode, task, k, additionalEqualities)
    try:
        b.run()
        return b.results(raw=raw)
    finally:
        b.close()


def align_batch_oneshot(queries, target, targets=None, mode="HW", task="distance", k=-1, additionalEqualities=None):
    """The one-shot C entry points edlibAlignBatchSharedTarget / edlibAlignBatchPairs (pointer arrays in,
    EdlibAlignResult[] out; shards over EDLIB_AMD_DEVICES).  Returns raw result dicts."""
    L = lib()
    n = len(queries)
    cfg, keep = _make_config(mode, task, k, additionalEqualities)
    qs = [bytes(q) for q in queries]
    qarr = (C.c_char_p * max(n, 1))(*qs)
    qlen = (C.c_int * max(n, 1))(*[len(q) for q in qs])
    res = (AlignResult * max(n, 1))()
    if targets is None:
        rc = L.edlibAlignBatchSharedTarget(qarr, qlen, n, bytes(target), len(target), cfg, res)
    else:
        ts = [bytes(t) for t in targets]
        tarr = (C.c_char_p * max(n, 1))(*ts)
        tlen = (C.c_int * max(n, 1))(*[len(t) for t in ts])
        rc = L.edlibAlignBatchPairs(qarr, qlen, tarr, tlen, n, cfg, res)
    if rc != 0:
        raise RuntimeError("edlib_amd: one-shot batch failed: " + last_error())
    out = []
    for i in range(n):
        out.append(_raw_result(res[i]))
        L.edlibFreeAlignResult(res[i])
    return out


def align_pairs(queries, targets, mode="NW", task="distance", k=-1, additionalEqualities=None, raw=False):
    """[align(q, t, ...) for q, t in zip(queries, targets)] in one device batch."""
    b = PairBatch(queries, targets, mode, task, k, additionalEqualities)
    try:
        b.run()
        return b.results(raw=raw)
    finally:
        b.close()
