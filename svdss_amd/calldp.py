"""Host-side bindings of the DP seams of the reference's `call` stage over the C-ABI.

`ksw_extd2_global` stands for the ksw_extd2_sse call at /root/reference/caller.cpp:332-355
(consensus -> reference window, scoring constants of caller.cpp:333-337), `fuzz_ratio` for
rapidfuzz::fuzz::ratio at caller.cpp:456,458, `run_poa` for Caller::run_poa / abpoa_msa at caller.cpp:257-308.  Compute runs in libsvdss_hip.so (HIP); there
is no CPU fallback.
"""
import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import check, lib
from .pingpong import pack_reads

# caller.hpp:25-37 _char26_table: A/a->0 C/c->1 G/g->2 T/t->3, everything else 4
CHAR26 = np.full(256, 4, dtype=np.uint8)
for _ch, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3)):
    CHAR26[ord(_ch)] = _v
    CHAR26[ord(_ch.lower())] = _v

# caller.cpp:333-337
SC_MCH, SC_MIS, GAPO, GAPE, GAPO2, GAPE2 = 1, -9, 16, 2, 41, 1
_a, _b = SC_MCH, SC_MIS
KSW_MAT = np.array([_a, _b, _b, _b, 0, _b, _a, _b, _b, 0, _b, _b, _a, _b, 0, _b, _b, _b, _a, 0, 0, 0, 0, 0, 0],
                   dtype=np.int8)


def encode26(s) -> np.ndarray:
    if isinstance(s, str):
        s = s.encode()
    if isinstance(s, (bytes, bytearray)):
        return CHAR26[np.frombuffer(bytes(s), dtype=np.uint8)]
    return np.ascontiguousarray(s, dtype=np.uint8)


def cigar_string(ops: Sequence[int]) -> str:
    """caller.cpp:352-355: to_string(c >> 4) + "MID"[c & 0xf]."""
    return "".join(f"{int(c) >> 4}{'MID'[int(c) & 0xf]}" for c in ops)


def ksw_extd2_global(queries: Sequence, targets: Sequence, device: int = 0, mat: np.ndarray = KSW_MAT,
                     gapo: int = GAPO, gape: int = GAPE, gapo2: int = GAPO2, gape2: int = GAPE2
                     ) -> Tuple[np.ndarray, List[np.ndarray], dict]:
    """Batch of global dual-affine alignments; returns (scores, [cigar ops per pair], stats)."""
    q, qo = pack_reads([encode26(x) for x in queries])
    t, to = pack_reads([encode26(x) for x in targets])
    assert len(qo) == len(to)
    n = len(qo) - 1
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    m = int(round(len(mat) ** 0.5))
    h = C.c_void_p()
    try:
        check(lib.svdss_align_global_batch(q.ctypes.data, qo.ctypes.data, t.ctypes.data, to.ctypes.data, n, m,
                                           mat.ctypes.data, gapo, gape, gapo2, gape2, device, C.byref(h)),
              "svdss_align_global_batch")
        scores = np.zeros(n, dtype=np.int32)
        nc = np.zeros(n, dtype=np.int64)
        cg = np.zeros(lib.svdss_aln_batch_total_cigar(h), dtype=np.uint32)
        check(lib.svdss_aln_batch_fetch(h, scores.ctypes.data, nc.ctypes.data, cg.ctypes.data),
              "svdss_aln_batch_fetch")
        stats = {"cells": lib.svdss_aln_batch_cells(h), "kernel_ms": lib.svdss_aln_batch_kernel_ms(h)}
    finally:
        if h:
            lib.svdss_aln_batch_free(h)
    out, o = [], 0
    for k in nc.tolist():
        out.append(cg[o:o + k].copy())
        o += k
    return scores, out, stats


def fuzz_ratio(a_list: Sequence, b_list: Sequence, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """rapidfuzz::fuzz::ratio for each (a, b) pair of byte strings; returns (ratio float64, lcs int64)."""
    def raw(x):
        if isinstance(x, str):
            x = x.encode()
        return np.frombuffer(bytes(x), dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else \
            np.ascontiguousarray(x, dtype=np.uint8)
    a, ao = pack_reads([raw(x) for x in a_list])
    b, bo = pack_reads([raw(x) for x in b_list])
    n = len(ao) - 1
    ratio = np.zeros(n, dtype=np.float64)
    lcs = np.zeros(n, dtype=np.int64)
    check(lib.svdss_indel_ratio_batch(a.ctypes.data, ao.ctypes.data, b.ctypes.data, bo.ctypes.data, n, device,
                                      ratio.ctypes.data, lcs.ctypes.data), "svdss_indel_ratio_batch")
    return ratio, lcs


def run_poa(clusters: Sequence[Sequence], device: int = 0):
    """Caller::run_poa (caller.cpp:257-308) for a batch of sub-clusters: clusters[c] = list of reads
    (str/bytes, or uint8 arrays already in 0..4 code) in BAM iteration order.  Returns
    ([consensus str over ACGTN per cluster], stats)."""
    seqs = [encode26(s) for cl in clusters for s in cl]
    flat, seq_off = pack_reads(seqs)
    cluster_off = np.zeros(len(clusters) + 1, dtype=np.int64)
    cluster_off[1:] = np.cumsum([len(cl) for cl in clusters])
    h = C.c_void_p()
    try:
        check(lib.svdss_poa_consensus_batch(flat.ctypes.data, seq_off.ctypes.data, cluster_off.ctypes.data,
                                            len(clusters), device, C.byref(h)), "svdss_poa_consensus_batch")
        lens = np.zeros(len(clusters), dtype=np.int64)
        cons = np.zeros(lib.svdss_poa_batch_total(h), dtype=np.uint8)
        check(lib.svdss_poa_batch_fetch(h, lens.ctypes.data, cons.ctypes.data), "svdss_poa_batch_fetch")
        stats = {"cells": lib.svdss_poa_batch_cells(h), "kernel_ms": lib.svdss_poa_batch_kernel_ms(h),
                 "hbm": lib.svdss_poa_batch_hbm(h),
                 "quad_back": lib.svdss_poa_batch_quad_back(h)}
    finally:
        if h:
            lib.svdss_poa_batch_free(h)
    out, o = [], 0
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for l in lens.tolist():
        out.append(bytes(letters[cons[o:o + l]]).decode())   # caller.cpp:297
        o += l
    return out, stats
