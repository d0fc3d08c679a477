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


# ------------------------------------------------------------------ native thread pool over a batch

class _RefPoolOut(C.Structure):     # ref_pool.cpp: struct RefPoolOut
    _fields_ = [("n", C.c_int), ("threads", C.c_int), ("wall_seconds", C.c_double),
                ("status", C.POINTER(C.c_int)), ("editDistance", C.POINTER(C.c_int)),
                ("numLocations", C.POINTER(C.c_int)), ("alphabetLength", C.POINTER(C.c_int)),
                ("alignmentLength", C.POINTER(C.c_int)),
                ("hasEnds", C.POINTER(C.c_ubyte)), ("hasStarts", C.POINTER(C.c_ubyte)), ("hasAlignment", C.POINTER(C.c_ubyte)),
                ("locOff", C.POINTER(C.c_longlong)), ("ends", C.POINTER(C.c_int)), ("starts", C.POINTER(C.c_int)),
                ("alnOff", C.POINTER(C.c_longlong)), ("alignment", C.POINTER(C.c_ubyte)),
                ("cigExtOff", C.POINTER(C.c_longlong)), ("cigStdOff", C.POINTER(C.c_longlong)),
                ("cigExt", C.POINTER(C.c_char)), ("cigStd", C.POINTER(C.c_char)),
                ("error", C.c_char * 256)]


_pool_lib = None


def _pool():
    global _pool_lib
    if _pool_lib is None:
        path = os.path.join(HERE, "libref_pool.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ref_pool_run.restype = C.POINTER(_RefPoolOut)
        L.ref_pool_run.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
        L.ref_pool_free.argtypes = [C.POINTER(_RefPoolOut)]
        L.ref_pool_physical_cores.restype = C.c_int
        _pool_lib = L
    return _pool_lib


def physical_cores():
    return int(_pool().ref_pool_physical_cores())


def cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota (cpu.max / cfs_quota_us) when there is one,
    capped by the affinity mask.  The GPU boxes show 256 logical CPUs behind a 16-CPU quota: 256 threads then
    run at 1/16 speed each and measure the throttle, not the reference."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def checker_library():
    """(path, kind) of the CPU implementation the pool should drive: the compiled reference where it
    travelled ("reference"), else the C99 restatement ("port")."""
    p = os.path.join(HERE, "_ref", "libedlib_ref.so")
    if os.path.exists(p):
        return p, "reference"
    load_oracle()
    return os.path.join(HERE, "liboracle_edlib.so"), "port"


def pool_align(qpool, qoff, tpool, toff, shared, mode, task, k=-1, select=None, threads=0, eq_pairs=None,
               want_cigar=False, libpath=None):
    """Run edlibAlign over a packed batch on a native std::thread pool (ref_pool.cpp).

    qpool / tpool: contiguous uint8 numpy arrays, qoff / toff: int64 offsets (toff has 2 entries when
    `shared`).  `select`: optional int32 array of unit indices (the bounded sample of a big batch).
    Returns a dict of flat numpy arrays (same layout as edlib_amd's flat results) plus wall_seconds / threads."""
    import numpy as np
    L = _pool()
    if libpath is None:
        libpath, _ = checker_library()
    qpool = np.ascontiguousarray(qpool, dtype=np.uint8); tpool = np.ascontiguousarray(tpool, dtype=np.uint8)
    qoff = np.ascontiguousarray(qoff, dtype=np.int64); toff = np.ascontiguousarray(toff, dtype=np.int64)
    n = len(qoff) - 1
    sel = None if select is None else np.ascontiguousarray(select, dtype=np.int32)
    eqb = b"".join((a if isinstance(a, bytes) else a.encode("latin-1"))[:1] + (b if isinstance(b, bytes) else b.encode("latin-1"))[:1]
                   for a, b in (eq_pairs or []))
    p = L.ref_pool_run(libpath.encode(), threads, qpool.ctypes.data, qoff.ctypes.data, tpool.ctypes.data, toff.ctypes.data,
                       1 if shared else 0, n, None if sel is None else sel.ctypes.data, 0 if sel is None else len(sel),
                       k, MODES[mode] if isinstance(mode, str) else mode, TASKS[task] if isinstance(task, str) else task,
                       eqb, len(eqb) // 2, 1 if want_cigar else 0)
    if not p:
        raise MemoryError("ref_pool_run")
    o = p.contents
    try:
        if o.error:
            raise RuntimeError("ref_pool: " + o.error.decode())
        cnt = o.n

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)
        loc = arr(o.locOff, cnt + 1, np.int64); aln = arr(o.alnOff, cnt + 1, np.int64)
        ce = arr(o.cigExtOff, cnt + 1, np.int64); cs = arr(o.cigStdOff, cnt + 1, np.int64)
        out = {"n": cnt, "threads": o.threads, "wall_seconds": o.wall_seconds,
               "status": arr(o.status, cnt, np.int32), "editDistance": arr(o.editDistance, cnt, np.int32),
               "numLocations": arr(o.numLocations, cnt, np.int32), "alphabetLength": arr(o.alphabetLength, cnt, np.int32),
               "alignmentLength": arr(o.alignmentLength, cnt, np.int32),
               "hasEnds": arr(o.hasEnds, cnt, np.uint8), "hasStarts": arr(o.hasStarts, cnt, np.uint8),
               "hasAlignment": arr(o.hasAlignment, cnt, np.uint8),
               "locOff": loc, "ends": arr(o.ends, int(loc[-1]), np.int32), "starts": arr(o.starts, int(loc[-1]), np.int32),
               "alnOff": aln, "alignment": arr(o.alignment, int(aln[-1]), np.uint8),
               "cigExtOff": ce, "cigStdOff": cs,
               "cigExt": C.string_at(o.cigExt, int(ce[-1])), "cigStd": C.string_at(o.cigStd, int(cs[-1]))}
    finally:
        L.ref_pool_free(p)
    return out
