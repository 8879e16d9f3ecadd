"""Call-side DP kernels (global dual-affine alignment with traceback; LCS ratio) against the
oracle: scores, CIGARs and ratios bit-exact."""
import os

import numpy as np
import pytest

from tests.mirror import caller
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
MAT = caller.KSW_MAT


def _pairs(seed, n, max_len, sv=True):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for k in range(n):
        tl = int(rng.integers(1, max_len))
        t = rng.integers(0, 5 if k % 7 == 0 else 4, size=tl).astype(np.uint8)
        q = t.copy()
        for _ in range(int(rng.integers(0, 5))):
            q[int(rng.integers(0, len(q)))] = int(rng.integers(0, 4))
        if sv and tl > 60:
            at, ln = int(rng.integers(10, tl - 40)), int(rng.integers(1, min(tl // 3, 400)))
            if rng.random() < 0.5:
                q = np.concatenate([q[:at], rng.integers(0, 4, size=ln).astype(np.uint8), q[at:]])
            else:
                q = np.concatenate([q[:at], q[at + min(ln, tl - at - 5):]])
        qs.append(q)
        ts.append(t)
    return qs, ts


def test_alignment_matches_oracle_small_and_ragged():
    qs, ts = _pairs(1, 200, 300)
    qs += [np.zeros(0, np.uint8), np.array([0], np.uint8), np.array([1, 2, 3], np.uint8)]
    ts += [np.array([0, 1], np.uint8), np.array([2], np.uint8), np.zeros(0, np.uint8)]
    scores, cigars, stats = caller.ksw_extd2_global(qs, ts)
    for q, t, s, c in zip(qs, ts, scores.tolist(), cigars):
        es, ec = O.ksw_extd2_global(q, t, MAT)
        assert s == es
        assert c.tolist() == ec.tolist()
    assert stats["cells"] == sum(len(q) * len(t) for q, t in zip(qs, ts))


def test_alignment_matches_oracle_cluster_sizes():
    # sub-cluster shapes of SURVEY 2.3 K4: hundreds to thousands of bases per side, long indels
    qs, ts = _pairs(2, 24, 3000)
    scores, cigars, stats = caller.ksw_extd2_global(qs, ts)
    for q, t, s, c in zip(qs, ts, scores.tolist(), cigars):
        es, ec = O.ksw_extd2_global(q, t, MAT)
        assert s == es and c.tolist() == ec.tolist()
        assert O.cigar_score(q, t, MAT, c) == s
    assert stats["kernel_ms"] > 0


def test_left_alignment_and_strings():
    unit = np.array([0, 1, 2], dtype=np.uint8)
    left = np.array([3, 3, 1, 0, 2, 3, 1, 2, 0, 0], dtype=np.uint8)
    right = np.array([1, 3, 3, 0, 2, 1, 0, 3, 2, 2], dtype=np.uint8)
    t = np.concatenate([left, np.tile(unit, 30), right])
    q = np.concatenate([left, np.tile(unit, 10), right])
    scores, cigars, _ = caller.ksw_extd2_global([q, t, "ACGTNACGT"], [t, q, "ACGTAACGT"])
    assert caller.cigar_string(cigars[0]) == "10M60D40M"
    assert caller.cigar_string(cigars[1]) == "10M60I40M"
    assert scores[2] == 8 and caller.cigar_string(cigars[2]) == "9M"


def test_large_pair_properties():
    # 8 kb x 8 kb: the oracle's O(nm) DP still runs in seconds; also check co-optimality of the CIGAR
    rng = np.random.default_rng(5)
    t = rng.integers(0, 4, size=8000).astype(np.uint8)
    q = np.concatenate([t[:3000], rng.integers(0, 4, size=700).astype(np.uint8), t[3000:5000], t[5600:]])
    scores, cigars, stats = caller.ksw_extd2_global([q], [t])
    es, ec = O.ksw_extd2_global(q, t, MAT)
    assert scores[0] == es and cigars[0].tolist() == ec.tolist()
    ops = [(int(c) >> 4, "MID"[c & 0xf]) for c in cigars[0]]
    assert (700, "I") in ops and (600, "D") in ops


def test_fuzz_ratio_matches_oracle():
    rng = np.random.default_rng(7)
    a_list, b_list = [], []
    for k in range(300):
        la = int(rng.integers(0, 400))
        a = rng.integers(65, 69, size=la).astype(np.uint8)
        b = a.copy()
        for _ in range(int(rng.integers(0, 30))):
            if len(b):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(65, 69))
        if k % 5 == 0:
            b = rng.integers(65, 69, size=int(rng.integers(0, 400))).astype(np.uint8)
        a_list.append(bytes(a))
        b_list.append(bytes(b))
    a_list += [b"", b"ACGT", b"this is a test"]
    b_list += [b"", b"", b"this is a test!"]
    ratio, lcs = caller.fuzz_ratio(a_list, b_list)
    for a, b, r, l in zip(a_list, b_list, ratio.tolist(), lcs.tolist()):
        assert l == O.lcs(a, b)
        assert r == O.fuzz_ratio(a, b)      # same double operations, bit-identical
    assert ratio[-3] == 100.0 and ratio[-2] == 0.0
    # alleles of SV size (caller.cpp:456-458 compares REF/ALT of up to ~10 kb)
    a = bytes(rng.integers(65, 69, size=6000).astype(np.uint8))
    b = a[:2500] + a[2600:]
    ratio, lcs = caller.fuzz_ratio([a], [b])
    assert lcs[0] == O.lcs(a, b) and ratio[0] == O.fuzz_ratio(a, b) and ratio[0] > 70


def test_fuzz_ratio_bit_parallel_kernel():
    """batches the bit-parallel LCS kernel takes (at most 8 distinct symbols, shorter string <= 4,096): symbol codes 0..3
    as the bench passes them, ASCII alleles as `SVDSS call` does; word boundaries of the 64 x 64-bit vector, carries that
    run through every word (equal strings), empty strings, a text much longer than the pattern"""
    rng = np.random.default_rng(21)
    for alphabet in (np.arange(4, dtype=np.uint8), np.frombuffer(b"ACGTN", dtype=np.uint8), np.array([7, 200, 255], dtype=np.uint8)):
        a_list, b_list = [], []
        for la in (0, 1, 2, 63, 64, 65, 127, 128, 129, 1000, 4095, 4096):
            a = rng.choice(alphabet, size=la)
            for kind in range(4):
                if kind == 0:
                    b = a.copy()                                        # equal: the carry runs through all words
                elif kind == 1:
                    b = a.copy()
                    for _ in range(1 + la // 50):
                        if len(b):
                            b[int(rng.integers(0, len(b)))] = rng.choice(alphabet)
                    b = np.delete(b, slice(la // 3, la // 3 + la // 10))
                elif kind == 2:
                    b = rng.choice(alphabet, size=int(rng.integers(0, 3000)))
                else:
                    b = np.concatenate([rng.choice(alphabet, size=700), a, rng.choice(alphabet, size=900)])   # longer text
                a_list.append(bytes(a.astype(np.uint8)))
                b_list.append(bytes(b.astype(np.uint8)))
        a_list.append(bytes(np.full(4096, alphabet[0], dtype=np.uint8)))
        b_list.append(bytes(np.full(20000, alphabet[0], dtype=np.uint8)))
        ratio, lcs = caller.fuzz_ratio(a_list, b_list)
        for a, b, r, l in zip(a_list, b_list, ratio.tolist(), lcs.tolist()):
            assert l == O.lcs(a, b), (len(a), len(b))
            assert r == O.fuzz_ratio(a, b)
        os.environ["SVDSS_RATIO_DP"] = "1"                              # the anti-diagonal kernel on the same batch
        try:
            ratio2, lcs2 = caller.fuzz_ratio(a_list, b_list)
        finally:
            del os.environ["SVDSS_RATIO_DP"]
        assert (ratio2 == ratio).all() and (lcs2 == lcs).all()
