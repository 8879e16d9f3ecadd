"""ctypes wrapper of oracle/libsvdss_oracle.so (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (SVDSS_ORACLE_LIB: another build of the checker, e.g. the sanitized one of `make -C oracle san`)
_lib = C.CDLL(os.environ.get("SVDSS_ORACLE_LIB") or os.path.join(ROOT, "oracle", "libsvdss_oracle.so"))

_p, _i64 = C.c_void_p, C.c_int64
_lib.orc_fmd_build.restype = _p
_lib.orc_fmd_build.argtypes = [_p, _p, C.c_int]
_lib.orc_fmd_from_bwt.restype = _p
_lib.orc_fmd_from_bwt.argtypes = [_p, _i64]
_lib.orc_fmd_free.argtypes = [_p]
_lib.orc_fmd_n.restype = _i64
_lib.orc_fmd_n.argtypes = [_p]
_lib.orc_fmd_acc.argtypes = [_p, _p]
_lib.orc_fmd_bwt.restype = _p
_lib.orc_fmd_bwt.argtypes = [_p]
_lib.orc_build_text.restype = _p
_lib.orc_build_text.argtypes = [_p, _p, C.c_int, _p]
_lib.orc_ping_pong_search.restype = _i64
_lib.orc_ping_pong_search.argtypes = [_p, _p, C.c_int, C.c_int, _p, _p, _i64, _p]
_lib.orc_ping_pong_bruteforce.restype = _i64
_lib.orc_ping_pong_bruteforce.argtypes = [_p, _i64, _p, C.c_int, C.c_int, _p, _p, _i64, _p]
_lib.orc_assemble.restype = _i64
_lib.orc_assemble.argtypes = [_p, _p, _i64, _p, _p]
_lib.orc_search_batch.restype = _i64
_lib.orc_search_batch.argtypes = [_p, _p, _p, _i64, C.c_int, C.c_int, _p, _p, C.POINTER(_p), C.POINTER(_p)]
_lib.orc_free.argtypes = [_p]
_lib.orc_nt6_encode.argtypes = [C.c_char_p, _i64, _p]
_lib.orc_max_threads.restype = C.c_int
_lib.orc_terminator_reads.restype = _i64
_lib.orc_terminator_reads.argtypes = []
_lib.orc_fmd_set_intv.argtypes = [_p, C.c_int, _p]
_lib.orc_fmd_extend.argtypes = [_p, _p, _p, C.c_int]


def terminator_reads() -> int:
    """accesses at or behind a read's terminator P[l] by the forward loops of both restatements since the library was
    loaded (oracle/svdss_oracle.c: provably none)"""
    return _lib.orc_terminator_reads()


def nt6_encode(s: bytes) -> np.ndarray:
    out = np.empty(len(s), dtype=np.uint8)
    _lib.orc_nt6_encode(s, len(s), out.ctypes.data)
    return out


def _flat(contigs):
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    flat = np.ascontiguousarray(np.concatenate(contigs), dtype=np.uint8)
    return flat, lens


def build_text(contigs) -> np.ndarray:
    flat, lens = _flat(contigs)
    n = _i64()
    p = _lib.orc_build_text(flat.ctypes.data, lens.ctypes.data, len(contigs), C.byref(n))
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()
    _lib.orc_free(p)
    return out


class OracleFMD:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def build(cls, contigs):
        flat, lens = _flat(contigs)
        return cls(_lib.orc_fmd_build(flat.ctypes.data, lens.ctypes.data, len(contigs)))

    @classmethod
    def from_bwt(cls, bwt: np.ndarray):
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        return cls(_lib.orc_fmd_from_bwt(bwt.ctypes.data, len(bwt)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_fmd_free(self.h)
            self.h = None

    @property
    def n(self):
        return _lib.orc_fmd_n(self.h)

    @property
    def acc(self):
        a = np.zeros(7, dtype=np.int64)
        _lib.orc_fmd_acc(self.h, a.ctypes.data)
        return a

    def bwt(self):
        p = _lib.orc_fmd_bwt(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n,)).copy()

    def count(self, pattern: np.ndarray) -> int:
        """interval size after set_intv + backward extends (ping_pong.cpp:12-22 without the stop)."""
        ik = (_i64 * 4)()
        ok = (_i64 * 24)()
        _lib.orc_fmd_set_intv(self.h, int(pattern[-1]), ik)
        for c in pattern[-2::-1]:
            if ik[2] == 0:
                break
            _lib.orc_fmd_extend(self.h, ik, ok, 1)
            for j in range(4):
                ik[j] = ok[4 * int(c) + j]
        return ik[2]

    def ping_pong_search(self, read: np.ndarray, overlap: int = -1):
        """orc_ping_pong_search on one read -> ([(qs,l)...] in push order, n_ext)."""
        l = len(read)
        P = np.zeros(l + 1, dtype=np.uint8)
        P[:l] = read
        cap = l + 1
        qs = np.zeros(cap, dtype=np.int32)
        ln = np.zeros(cap, dtype=np.int32)
        ext = _i64()
        n = _lib.orc_ping_pong_search(self.h, P.ctypes.data, l, overlap, qs.ctypes.data, ln.ctypes.data,
                                      cap, C.byref(ext))
        assert n >= 0
        return list(zip(qs[:n].tolist(), ln[:n].tolist())), ext.value

    def search_batch(self, flat: np.ndarray, offsets: np.ndarray, assemble: bool, threads: int = 0):
        """orc_search_batch: reads WITHOUT terminators in, (counts, qs, len, n_ext) out."""
        n = len(offsets) - 1
        # the restated loop reads P[l] == 0 in principle (ping_pong.cpp:94): add terminators
        lens = np.diff(offsets)
        toff = offsets + np.arange(n + 1, dtype=np.int64)
        tflat = np.zeros(int(toff[-1]), dtype=np.uint8)
        if n:
            idx = np.arange(int(offsets[-1]), dtype=np.int64) + np.repeat(np.arange(n, dtype=np.int64), lens)
            tflat[idx] = flat
        counts = np.zeros(n, dtype=np.int64)
        n_ext = np.zeros(n, dtype=np.int64)
        pq, pl = _p(), _p()
        if threads <= 0:
            threads = _lib.orc_max_threads()
        total = _lib.orc_search_batch(self.h, tflat.ctypes.data, toff.ctypes.data, n, int(assemble), threads,
                                      counts.ctypes.data, n_ext.ctypes.data, C.byref(pq), C.byref(pl))
        qs = np.ctypeslib.as_array(C.cast(pq, C.POINTER(C.c_int32)), shape=(max(total, 1),))[:total].copy()
        ln = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_int32)), shape=(max(total, 1),))[:total].copy()
        _lib.orc_free(pq)
        _lib.orc_free(pl)
        return counts, qs, ln, n_ext


def ping_pong_bruteforce(text: np.ndarray, read: np.ndarray, overlap: int = -1):
    l = len(read)
    P = np.zeros(l + 1, dtype=np.uint8)
    P[:l] = read
    cap = l + 1
    qs = np.zeros(cap, dtype=np.int32)
    ln = np.zeros(cap, dtype=np.int32)
    ext = _i64()
    text = np.ascontiguousarray(text, dtype=np.uint8)
    n = _lib.orc_ping_pong_bruteforce(text.ctypes.data, len(text), P.ctypes.data, l, overlap,
                                      qs.ctypes.data, ln.ctypes.data, cap, C.byref(ext))
    assert n >= 0
    return list(zip(qs[:n].tolist(), ln[:n].tolist())), ext.value


def assemble(sfs):
    n = len(sfs)
    qs = np.array([s[0] for s in sfs], dtype=np.int32)
    ln = np.array([s[1] for s in sfs], dtype=np.int32)
    oq = np.zeros(max(n, 1), dtype=np.int32)
    ol = np.zeros(max(n, 1), dtype=np.int32)
    m = _lib.orc_assemble(qs.ctypes.data, ln.ctypes.data, n, oq.ctypes.data, ol.ctypes.data)
    return list(zip(oq[:m].tolist(), ol[:m].tolist()))


def max_threads():
    return _lib.orc_max_threads()


# ---- call-side DP (oracle/svdss_oracle_call.c) -----------------------------
_i32 = C.c_int32
_lib.orc_ksw_extd2_global.restype = _i64
_lib.orc_ksw_extd2_global.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _p, _p, _i64]
_lib.orc_global_score_general.restype = _i32
_lib.orc_global_score_general.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.orc_cigar_score.restype = _i32
_lib.orc_cigar_score.argtypes = [_p, C.c_int, _p, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _i64]
_lib.orc_lcs.restype = _i64
_lib.orc_lcs.argtypes = [_p, _i64, _p, _i64]
_lib.orc_fuzz_ratio.restype = C.c_double
_lib.orc_fuzz_ratio.argtypes = [_p, _i64, _p, _i64]


def ksw_extd2_global(query, target, mat, q=16, e=2, q2=41, e2=1):
    """-> (score, cigar ops uint32 array)."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    m = int(round(len(mat) ** 0.5))
    cap = len(query) + len(target) + 2
    cg = np.zeros(cap, dtype=np.uint32)
    sc = _i32()
    n = _lib.orc_ksw_extd2_global(query.ctypes.data, len(query), target.ctypes.data, len(target), m,
                                  mat.ctypes.data, q, e, q2, e2, C.byref(sc), cg.ctypes.data, cap)
    assert n >= 0
    return sc.value, cg[:n].copy()


def global_score_general(query, target, mat, q=16, e=2, q2=41, e2=1):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    m = int(round(len(mat) ** 0.5))
    return _lib.orc_global_score_general(query.ctypes.data, len(query), target.ctypes.data, len(target), m,
                                         mat.ctypes.data, q, e, q2, e2)


def cigar_score(query, target, mat, cigar, q=16, e=2, q2=41, e2=1):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    m = int(round(len(mat) ** 0.5))
    return _lib.orc_cigar_score(query.ctypes.data, len(query), target.ctypes.data, len(target), m,
                                mat.ctypes.data, q, e, q2, e2, cigar.ctypes.data, len(cigar))


def lcs(a: bytes, b: bytes) -> int:
    a = np.frombuffer(a, dtype=np.uint8)
    b = np.frombuffer(b, dtype=np.uint8)
    return _lib.orc_lcs(a.ctypes.data, len(a), b.ctypes.data, len(b))


def fuzz_ratio(a: bytes, b: bytes) -> float:
    a = np.frombuffer(a, dtype=np.uint8)
    b = np.frombuffer(b, dtype=np.uint8)
    return _lib.orc_fuzz_ratio(a.ctypes.data, len(a), b.ctypes.data, len(b))


# ---- POA consensus (oracle/svdss_oracle_poa.c) ------------------------------
_lib.orc_poa_consensus.restype = _i64
_lib.orc_poa_consensus.argtypes = [_p, _p, C.c_int, _p, _i64]


def poa_consensus(seqs) -> np.ndarray:
    """seqs: list of uint8 arrays over 0..4 (ACGTN).  Returns the consensus symbols."""
    n = len(seqs)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    flat = np.ascontiguousarray(np.concatenate(seqs) if n and offs[-1] else np.zeros(0, np.uint8), dtype=np.uint8)
    cap = int(offs[-1]) + 8
    out = np.zeros(cap, dtype=np.uint8)
    m = _lib.orc_poa_consensus(flat.ctypes.data, offs.ctypes.data, n, out.ctypes.data, cap)
    assert m >= 0
    return out[:m].copy()


# ---- the call-side DP of a list of sub-clusters on host threads (oracle/svdss_oracle_callbatch.c) ---------------
_lib.orc_call_batch.restype = C.c_int
_lib.orc_call_batch.argtypes = [_p, _p, _p, _i64, _p, _p, _p, C.c_int, _p, _p, _p, _p]


def call_batch(seqs, seq_off, cluster_off, refs, ref_off, mat, threads=0):
    """POA -> realignment of every sub-cluster (OpenMP over sub-clusters, caller.cpp:319-321), then the ratio of
    adjacent consensus pairs.  -> (cons_len int64[n], score int32[n], n_cigar int64[n], ratio float64[n - 1])."""
    n = len(cluster_off) - 1
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
    cluster_off = np.ascontiguousarray(cluster_off, dtype=np.int64)
    refs = np.ascontiguousarray(refs, dtype=np.uint8)
    ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    cons_len = np.zeros(n, np.int64)
    score = np.zeros(n, np.int32)
    n_cig = np.zeros(n, np.int64)
    ratio = np.zeros(max(0, n - 1), np.float64)
    if threads <= 0:
        threads = _lib.orc_max_threads()
    rc = _lib.orc_call_batch(seqs.ctypes.data, seq_off.ctypes.data, cluster_off.ctypes.data, n, refs.ctypes.data,
                             ref_off.ctypes.data, mat.ctypes.data, threads, cons_len.ctypes.data, score.ctypes.data,
                             n_cig.ctypes.data, ratio.ctypes.data)
    assert rc == 0
    return cons_len, score, n_cig, ratio
