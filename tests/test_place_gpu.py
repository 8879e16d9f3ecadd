"""K7: SFS placement on the GPU (svdss_place_sfs_batch, csrc/place.hip) against the Python restatement of
Clusterer::extend_alignment (svdss_amd/clusterer.py, itself pinned by the hand-worked cases of tests/test_clusterer.py)
on randomised alignments: CIGARs with matches, insertions, deletions, soft clips and skips, SFS lists in file order and
shuffled (the reference carries `last_pos` from one SFS to the next), references with low-complexity stretches so that
k-mer uniqueness -- and the fall-through when no 7-mer is unique -- decide the borders."""
import ctypes as C

import numpy as np
import pytest

from svdss_amd import synth
from svdss_amd._lib import check, lib
from tests.mirror.clusterer import Alignment, Clusterer

pytestmark = pytest.mark.gpu
OPS = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "=": 7, "X": 8}


_PARTS = [False]      # alternates: the chromosomes back to back in one buffer / one buffer each (svdss_ref_upload_parts)


def _place(chrom_seqs, alns, sfs_lists):
    seqs = np.frombuffer("".join(chrom_seqs).encode(), np.uint8)
    off = np.zeros(len(chrom_seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in chrom_seqs])
    h = C.c_void_p()
    _PARTS[0] = not _PARTS[0]
    if _PARTS[0]:
        bufs = [np.frombuffer(s.encode(), np.uint8) if len(s) else np.zeros(0, np.uint8) for s in chrom_seqs]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data if len(b) else None for b in bufs])
        lens = np.array([len(b) for b in bufs], np.int64)
        check(lib.svdss_ref_upload_parts(ptrs, lens.ctypes.data, len(bufs), 0, C.byref(h)), "svdss_ref_upload_parts")
    else:
        check(lib.svdss_ref_upload(seqs.ctypes.data, off.ctypes.data, len(chrom_seqs), 0, C.byref(h)), "svdss_ref_upload")
    try:
        tid = np.array([a.tid for a in alns], np.int32)
        pos = np.array([a.pos for a in alns], np.int32)
        cig = np.array([(l << 4) | op for a in alns for l, op in a.cigar], np.uint32)
        cig_off = np.zeros(len(alns) + 1, np.int64)
        cig_off[1:] = np.cumsum([len(a.cigar) for a in alns])
        qs = np.array([s[0] for ss in sfs_lists for s in ss], np.int32)
        ln = np.array([s[1] for ss in sfs_lists for s in ss], np.int32)
        sfs_off = np.zeros(len(alns) + 1, np.int64)
        sfs_off[1:] = np.cumsum([len(ss) for ss in sfs_lists])
        cnt = np.zeros(len(alns), np.int32)
        out = np.zeros(5 * max(1, len(qs)), np.int32)
        stats = np.zeros(4, np.int64)
        check(lib.svdss_place_sfs_batch(h, tid.ctypes.data, pos.ctypes.data, cig.ctypes.data, cig_off.ctypes.data,
                                        qs.ctypes.data, ln.ctypes.data, sfs_off.ctypes.data, len(alns), cnt.ctypes.data,
                                        out.ctypes.data, stats.ctypes.data), "svdss_place_sfs_batch")
    finally:
        lib.svdss_ref_free(h)
    res = []
    for i in range(len(alns)):
        o = 5 * int(sfs_off[i])
        res.append([tuple(out[o + 5 * j:o + 5 * j + 5].tolist()) for j in range(int(cnt[i]))])
    return res, stats.tolist()


def _random_case(rng, chroms, k):
    ci = int(rng.integers(0, len(chroms)))
    clen = len(chroms[ci])
    n_ops = int(rng.integers(1, 40))
    cigar, qlen, rlen = [], 0, 0
    if rng.random() < 0.3:
        l = int(rng.integers(1, 60)); cigar.append((l, OPS["S"])); qlen += l
    for _ in range(n_ops):
        x = rng.random()
        if x < 0.55:
            l = int(rng.integers(1, 400)); op = "M" if rng.random() < 0.8 else ("=" if rng.random() < 0.5 else "X")
        elif x < 0.75:
            l = int(rng.integers(1, 80)); op = "I"
        elif x < 0.93:
            l = int(rng.integers(1, 80)); op = "D"
        elif x < 0.97:
            l = int(rng.integers(1, 300)); op = "N"
        else:
            l = 0; op = "M"                                    # zero-length operation
        cigar.append((l, OPS[op]))
        if op in "M=XI":
            qlen += l
        if op in "M=XDN":
            rlen += l
    if rng.random() < 0.3:
        l = int(rng.integers(1, 60)); cigar.append((l, OPS["S"])); qlen += l
    if rng.random() < 0.1:
        cigar.append((5, OPS["H"]))
    pos = int(rng.integers(0, max(1, clen - rlen - 1)))
    n_sfs = int(rng.integers(0, 9))
    sfs = []
    for _ in range(n_sfs):
        q = int(rng.integers(0, max(1, qlen)))
        sfs.append((q, int(rng.integers(1, 120)), int(rng.integers(0, 3))))
    if rng.random() < 0.7:
        sfs.sort()                                            # .sfs file order (ascending); otherwise as --noassemble leaves it
    name = f"r{k}"
    return Alignment(name, 0, ci, pos, 60, cigar, "A" * qlen), sfs


def test_placement_matches_the_host_restatement():
    rng = np.random.default_rng(7)
    chroms = []
    for n in (30000, 12000):
        c = rng.integers(1, 5, size=n).astype(np.uint8)
        for _ in range(6):                                     # low-complexity stretches: no unique 7-mer in a flank
            a = int(rng.integers(0, n - 700))
            c[a:a + 600] = np.tile(rng.integers(1, 5, size=int(rng.integers(1, 4))).astype(np.uint8), 600)[:600]
        chroms.append(synth.to_ascii(c))
    names = ["c0", "c1"]
    alns, lists = [], []
    for k in range(3000):
        a, sfs = _random_case(rng, chroms, k)
        alns.append(a); lists.append(sfs)
    got, stats = _place(chroms, alns, lists)
    cl = Clusterer({a.qname: s for a, s in zip(alns, lists)}, dict(zip(names, chroms)), names, threads=1)
    n_ext = n_merged = 0
    for a, sfs, g in zip(alns, lists, got):
        want = cl.extend_alignment(a)
        assert [(x.rs, x.re, x.qs, x.qe) for x in want] == [t[:4] for t in g], (a.qname, a.pos, a.cigar, sfs)
        assert [x.htag for x in want] == [sfs[t[4]][2] for t in g]
        n_ext += len(g)
        n_merged += len(sfs) - len(g)
    assert stats == [cl.unplaced, cl.s_unplaced, cl.e_unplaced, cl.unknown]
    assert n_ext > 3000 and n_merged > 500 and stats[1] > 10 and stats[2] > 10
