"""ctypes loaders for the TEST-ONLY checkers (never imported by edlib_amd/).

  load_oracle()  -> the C99 restatement, oracle/liboracle_edlib.so
  load_ref()     -> the compiled, unmodified reference, oracle/_ref/libedlib_ref.so
                    (None when it has not been built / did not travel)

Both expose ``align(query, target, mode, task, k, eq_pairs) -> dict`` with the
fields of EdlibAlignResult (reference edlib/include/edlib.h:162-218) and
``cigar(ops, fmt)``.  Sequences are ``bytes``.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"NW": 0, "SHW": 1, "HW": 2}
TASKS = {"distance": 0, "locations": 1, "path": 2}


class EqPair(C.Structure):
    _fields_ = [("first", C.c_char), ("second", C.c_char)]


class AlignConfig(C.Structure):  # edlib.h:100-140
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int),
                ("additionalEqualities", C.POINTER(EqPair)),
                ("additionalEqualitiesLength", C.c_int)]


class AlignResult(C.Structure):  # edlib.h:162-218
    _fields_ = [("status", C.c_int), ("editDistance", C.c_int),
                ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)),
                ("numLocations", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)),
                ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]


def result_to_dict(r):
    n = r.numLocations
    return {
        "status": r.status,
        "editDistance": r.editDistance,
        "endLocations": [r.endLocations[i] for i in range(n)] if r.endLocations else None,
        "startLocations": [r.startLocations[i] for i in range(n)] if r.startLocations else None,
        "numLocations": n,
        "alignment": bytes(r.alignment[i] for i in range(r.alignmentLength)) if r.alignment else None,
        "alignmentLength": r.alignmentLength,
        "alphabetLength": r.alphabetLength,
    }


def _eq_array(eq_pairs):
    if not eq_pairs:
        return None, 0
    arr = (EqPair * len(eq_pairs))()
    for i, (a, b) in enumerate(eq_pairs):
        arr[i].first = a if isinstance(a, bytes) else a.encode("latin-1")
        arr[i].second = b if isinstance(b, bytes) else b.encode("latin-1")
    return arr, len(eq_pairs)


class _Oracle:
    """The C restatement."""
    name = "oracle"

    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.lib.oracle_align.restype = AlignResult
        self.lib.oracle_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                          C.c_int, C.c_int, C.c_int,
                                          C.POINTER(EqPair), C.c_int]
        self.lib.oracle_free_result.argtypes = [C.POINTER(AlignResult)]
        self.lib.oracle_cigar.restype = C.c_void_p
        self.lib.oracle_cigar.argtypes = [C.c_char_p, C.c_int, C.c_int]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def align(self, query, target, mode="NW", task="distance", k=-1, eq_pairs=None):
        arr, n = _eq_array(eq_pairs)
        r = self.lib.oracle_align(query, len(query), target, len(target), k,
                                  MODES[mode] if isinstance(mode, str) else mode,
                                  TASKS[task] if isinstance(task, str) else task, arr, n)
        d = result_to_dict(r)
        self.lib.oracle_free_result(C.byref(r))
        return d

    def cigar(self, ops, fmt=1):
        p = self.lib.oracle_cigar(bytes(ops), len(ops), fmt)
        if not p:
            return None
        s = C.string_at(p).decode()
        self.libc.free(p)
        return s


class _Ref:
    """The compiled reference (or any library with the edlib C ABI)."""
    name = "reference"

    def __init__(self, path):
        self.lib = C.CDLL(path, mode=os.RTLD_LOCAL if hasattr(os, "RTLD_LOCAL") else 0)
        self.lib.edlibAlign.restype = AlignResult
        self.lib.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, AlignConfig]
        self.lib.edlibFreeAlignResult.argtypes = [AlignResult]
        self.lib.edlibAlignmentToCigar.restype = C.c_void_p
        self.lib.edlibAlignmentToCigar.argtypes = [C.c_char_p, C.c_int, C.c_int]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def align(self, query, target, mode="NW", task="distance", k=-1, eq_pairs=None):
        arr, n = _eq_array(eq_pairs)
        cfg = AlignConfig(k, MODES[mode] if isinstance(mode, str) else mode,
                          TASKS[task] if isinstance(task, str) else task,
                          C.cast(arr, C.POINTER(EqPair)) if arr is not None else None, n)
        r = self.lib.edlibAlign(query, len(query), target, len(target), cfg)
        d = result_to_dict(r)
        self.lib.edlibFreeAlignResult(r)
        return d

    def cigar(self, ops, fmt=1):
        p = self.lib.edlibAlignmentToCigar(bytes(ops), len(ops), fmt)
        if not p:
            return None
        s = C.string_at(p).decode()
        self.libc.free(p)
        return s


def build(quiet=True):
    """Compile the restatement (and, where /root/reference exists, _ref)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def load_oracle():
    path = os.path.join(HERE, "liboracle_edlib.so")
    if not os.path.exists(path):
        build()
    return _Oracle(path)


def load_ref():
    path = os.path.join(HERE, "_ref", "libedlib_ref.so")
    if not os.path.exists(path):
        return None
    return _Ref(path)
