"""Deterministic synthetic sequence batches (SURVEY.md §8d shapes).

Everything is derived from a counter-based splitmix64 hash, so the same
(seed, shape) gives the same bytes on every machine and numpy version -- the
golden fixtures under tests/golden/ store only seeds plus the reference's
answers, and bench.py / the GPU tests regenerate the inputs.

The real inputs of BASELINE.json config 2 (E. coli DH1 + mason reads,
reference test_data/E_coli_DH1, a missing large blob) are replaced by i.i.d.
uniform ACGT and Illumina-like edits, as SURVEY.md §8d prescribes.
"""
import numpy as np

_U64 = np.uint64
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mix64(x):
    """splitmix64 finalizer; consumes (overwrites) the uint64 array x."""
    x += _U64(0x9E3779B97F4A7C15)
    t = x >> _U64(30); x ^= t; x *= _U64(0xBF58476D1CE4E5B9)
    np.right_shift(x, _U64(27), out=t); x ^= t; x *= _U64(0x94D049BB133111EB)
    np.right_shift(x, _U64(31), out=t); x ^= t
    return x


def rand_u64(seed, n, stream=0, offset=0):
    """values [offset, offset+n) of the deterministic 64-bit sequence (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _mix64(np.array([seed], dtype=_U64) * _U64(0xD1342543DE82EF95)
                      + _U64(stream) * _U64(0xA24BAED4963EE407))[0]
        x = np.arange(offset, offset + n, dtype=_U64)
        x += base
        return _mix64(x)


def rand_unit(seed, n, stream=0):
    """n deterministic doubles in [0,1)."""
    return (rand_u64(seed, n, stream) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def random_dna(seed, n, stream=0):
    """n i.i.d. uniform bases as a uint8 array of ASCII 'A','C','G','T'."""
    return _ACGT[(rand_u64(seed, n, stream) >> _U64(33)) % _U64(4)]


def random_symbols(seed, n, sigma, stream=0, base=65):
    return ((rand_u64(seed, n, stream) >> _U64(33)) % _U64(sigma)).astype(np.uint8) + np.uint8(base)


def _other_base(b, r):
    """a base different from b (ASCII), chosen by r in {0,1,2}."""
    idx = np.searchsorted(_ACGT, b)  # ACGT is sorted in ASCII
    return _ACGT[(idx + 1 + r) % 4]


def mutate(seq, seed, sub, ins, dele, stream=0):
    """Apply per-base substitution / insertion / deletion to one sequence.

    Returns (mutated uint8 array, number of edits applied)."""
    seq = np.asarray(seq, dtype=np.uint8)
    n = len(seq)
    u = rand_unit(seed, n, stream * 4 + 0)
    r = (rand_u64(seed, n, stream * 4 + 1) >> _U64(40)).astype(np.int64)
    is_del = u < dele
    is_ins = (u >= dele) & (u < dele + ins)
    is_sub = (u >= dele + ins) & (u < dele + ins + sub)
    out = seq.copy()
    out[is_sub] = _other_base(seq[is_sub], r[is_sub] % 3)
    # insertion: a random base is emitted before the original one
    counts = np.where(is_del, 0, np.where(is_ins, 2, 1))
    total = int(counts.sum())
    pos = np.cumsum(counts) - counts
    res = np.empty(total, dtype=np.uint8)
    keep = ~is_del
    res[(pos + counts - 1)[keep]] = out[keep]
    res[pos[is_ins]] = _ACGT[r[is_ins] % 4]
    return res, int(is_del.sum() + is_ins.sum() + is_sub.sum())


def illumina_reads(target, n, m=150, seed=12346, sub=0.01, ins=0.0005, dele=0.0005,
                   frac_random=0.05, chunk=32768):
    """n reads of exactly m bases drawn from `target` (uint8 ASCII array).

    Read i copies m bases at a uniform start, gets per-base substitutions
    (rate `sub`), insertions and deletions (rates `ins`, `dele`), is padded from
    the genome / truncated back to exactly m, and with probability
    `frac_random` is replaced by an unrelated uniform read.

    Returns dict(reads=[n,m] uint8, start=[n] int64, edits=[n] int32 (upper
    bound on the planted edit count), random=[n] bool, indels=[n] int32 (planted insertions +
    deletions: a read with none and not `random` is its genome window with exactly `edits`
    substitutions, so it ends at start + m - 1 with that many mismatches)).  Generated in chunks
    of `chunk` reads; the result does not depend on the chunk size.
    """
    target = np.asarray(target, dtype=np.uint8)
    T = len(target)
    pad = 16
    assert T >= m + pad
    start = (rand_u64(seed, n, 1) % _U64(T - m - pad + 1)).astype(np.int64)
    is_random = rand_unit(seed, n, 4) < frac_random
    reads = np.empty((n, m), dtype=np.uint8)
    edits = np.empty(n, dtype=np.int32)
    indels = np.zeros(n, dtype=np.int32)
    th = [int(round(v * (1 << 24))) for v in (sub, sub + ins, sub + ins + dele)]
    cols = np.arange(m + pad)[None, :]
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        win = target[start[a:b, None] + cols]                    # [c, m+pad]
        x = rand_u64(seed, (b - a) * m, 2, offset=a * m).reshape(b - a, m)   # one draw per base:
        u = (x >> _U64(40)).astype(np.uint32)                    # top 24 bits pick the event,
        r = (x & _U64(0xFFFF)).astype(np.int32)                  # low 16 bits pick the base
        is_sub = u < th[0]
        is_ins = (u >= th[0]) & (u < th[1])
        is_del = (u >= th[1]) & (u < th[2])
        rd = win[:, :m].copy()
        rd[is_sub] = _other_base(rd[is_sub], r[is_sub] % 3)
        ed = is_sub.sum(axis=1).astype(np.int32)
        indel_rows = np.nonzero((is_ins | is_del).any(axis=1))[0]
        if len(indel_rows):                   # ~15 % of reads at the default rates
            k = len(indel_rows)
            src = np.concatenate([rd[indel_rows], win[indel_rows, m:]], axis=1)   # [k, m+pad]
            dmask = np.zeros((k, m + pad), dtype=bool); dmask[:, :m] = is_del[indel_rows]
            imask = np.zeros((k, m + pad), dtype=bool); imask[:, :m] = is_ins[indel_rows]
            counts = np.where(dmask, 0, np.where(imask, 2, 1))
            pos = np.cumsum(counts, axis=1) - counts        # output index of first emitted base
            rows = np.broadcast_to(np.arange(k)[:, None], pos.shape)
            out = np.zeros((k, m), dtype=np.uint8)
            last = pos + counts - 1                         # where the original base lands
            keep = (~dmask) & (last < m)
            out[rows[keep], last[keep]] = src[keep]
            insk = imask & (pos < m)
            rr = np.zeros((k, m + pad), dtype=np.int32); rr[:, :m] = r[indel_rows]
            out[rows[insk], pos[insk]] = _ACGT[rr[insk] % 4]
            rd[indel_rows] = out
            # an indel shifts the tail against the genome: count it generously
            ed[indel_rows] = (is_sub[indel_rows].sum(axis=1)
                              + 2 * (is_ins[indel_rows].sum(axis=1) + is_del[indel_rows].sum(axis=1)))
        reads[a:b] = rd
        edits[a:b] = ed
        indels[a:b] = (is_ins | is_del).sum(axis=1)
    nr = int(is_random.sum())
    if nr:
        reads[is_random] = random_dna(seed, nr * m, 5).reshape(nr, m)
    return {"reads": reads, "start": start, "edits": edits, "random": is_random, "indels": indels}


def _pair(args):
    i, length, seed, sub, ins, dele = args
    t = random_dna(seed, length, stream=1000 + 2 * i)
    q, _ = mutate(t, seed, sub, ins, dele, stream=1001 + 2 * i)
    if len(q) == 0:
        q = t[:1].copy()
    return q, t


def _pair_chunk(args):
    lo, hi, length, seed, sub, ins, dele = args
    return [_pair((i, length, seed, sub, ins, dele)) for i in range(lo, hi)]


def mutated_pairs(n, length, seed, sub, ins, dele, workers=1, first=0):
    """n (query, target) pairs: target = `length` uniform bases, query = target
    with edits.  Returns two lists of uint8 arrays.
    Pair i depends on (seed, first + i) only, so `workers` > 1 generates chunks in forked processes
    with the same result."""
    if workers > 1 and n >= 4 * workers:
        import multiprocessing as mp
        step = (n + 4 * workers - 1) // (4 * workers)
        jobs = [(first + a, first + min(n, a + step), length, seed, sub, ins, dele) for a in range(0, n, step)]
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_pair_chunk, jobs)
        pairs = [p for part in parts for p in part]
    else:
        pairs = _pair_chunk((first, first + n, length, seed, sub, ins, dele))
    return [p[0] for p in pairs], [p[1] for p in pairs]


def masked_genome(seed, n, frac_n=0.01, frac_lower=0.1, iupac=False):
    """uniform ACGT with runs of N (~50 long), soft-masked lower-case stretches (~200 long) and, with `iupac`,
    scattered IUPAC codes: the alphabets real reference genomes have (5 to 16 distinct bytes)."""
    g = random_dna(seed, n)
    u = rand_unit(seed, n, 11)
    nrun = np.zeros(n, dtype=bool); lrun = np.zeros(n, dtype=bool)
    for start in np.nonzero(u < frac_n / 50.0)[0]:
        nrun[start:start + 50] = True
    for start in np.nonzero((u >= 0.5) & (u < 0.5 + frac_lower / 200.0))[0]:
        lrun[start:start + 200] = True
    g = g.copy()
    g[lrun] = g[lrun] + 32                     # lower case
    g[nrun] = ord("N")
    if iupac:
        codes = np.frombuffer(b"RYKMSW", dtype=np.uint8)
        at = np.nonzero(rand_unit(seed, n, 12) < 0.002)[0]
        g[at] = codes[(rand_u64(seed, len(at), 13) >> _U64(20)) % _U64(len(codes))]
    return g


def window_reads(target, n, m, seed, sub=0.01):
    """n reads = m-byte windows of `target` (whatever bytes it holds) with substitutions by uniform ACGT."""
    target = np.asarray(target, dtype=np.uint8)
    start = (rand_u64(seed, n, 1) % _U64(len(target) - m + 1)).astype(np.int64)
    reads = target[start[:, None] + np.arange(m)[None, :]].copy()
    x = rand_u64(seed, n * m, 2).reshape(n, m)
    hit = (x >> _U64(40)).astype(np.uint32) < int(round(sub * (1 << 24)))
    reads[hit] = _ACGT[(x[hit] & _U64(3)).astype(np.int64)]
    return reads, start
