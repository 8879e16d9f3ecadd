"""Oracle of the call-side DP seams against its pins: the ksw2-style global alignment score
against an independent general-gap DP, the CIGAR against the score it implies and ksw2's
left-alignment convention, the ratio against the closed form of SURVEY App. B.4."""
import numpy as np
import pytest

from tests.mirror import caller
from tests import oracle_lib as O

MAT = caller.KSW_MAT


def _mutate(rng, s, n_sub=3, indels=((5, 40),)):
    s = s.copy()
    for _ in range(n_sub):
        s[int(rng.integers(0, len(s)))] = int(rng.integers(0, 4))
    for at_frac, ln in indels:
        at = int(len(s) * at_frac / 10)
        if rng.random() < 0.5:
            s = np.concatenate([s[:at], rng.integers(0, 4, size=ln).astype(np.uint8), s[at:]])
        else:
            s = np.concatenate([s[:at], s[at + ln:]])
    return s


@pytest.mark.parametrize("seed", range(6))
def test_score_is_the_optimum_and_cigar_attains_it(seed):
    rng = np.random.default_rng(seed)
    for _ in range(40):
        tl = int(rng.integers(1, 70))
        t = rng.integers(0, 5 if rng.random() < 0.2 else 4, size=tl).astype(np.uint8)
        q = _mutate(rng, t, n_sub=int(rng.integers(0, 4)), indels=((int(rng.integers(1, 9)), int(rng.integers(1, 30))),)) \
            if tl > 35 else rng.integers(0, 4, size=int(rng.integers(1, 50))).astype(np.uint8)
        if len(q) == 0:
            continue
        sc, cg = O.ksw_extd2_global(q, t, MAT)
        assert sc == O.global_score_general(q, t, MAT)
        assert O.cigar_score(q, t, MAT, cg) == sc
        # CIGAR consumes exactly the two sequences and never has two adjacent ops of one kind
        assert sum(int(c) >> 4 for c in cg if (c & 0xf) in (0, 1)) == len(q)
        assert sum(int(c) >> 4 for c in cg if (c & 0xf) in (0, 2)) == len(t)
        assert all((a & 0xf) != (b & 0xf) for a, b in zip(cg, cg[1:]))


def test_known_alignments():
    A, Cc, G, T = 0, 1, 2, 3
    t = np.array([A, Cc, G, T] * 10, dtype=np.uint8)
    sc, cg = O.ksw_extd2_global(t, t, MAT)
    assert sc == 40 and caller.cigar_string(cg) == "40M"
    # 50-bp deletion from the query: one D of 50, cost min(16+2*50, 41+50) = 91
    rng = np.random.default_rng(3)
    t = rng.integers(0, 4, size=300).astype(np.uint8)
    q = np.concatenate([t[:100], t[150:]])
    sc, cg = O.ksw_extd2_global(q, t, MAT)
    assert sc == 250 - 91
    assert caller.cigar_string(cg) in ("100M50D150M",) or (len(cg) == 3 and (cg[1] & 0xf) == 2 and (cg[1] >> 4) == 50)
    # insertion of 60 in the query
    q = np.concatenate([t[:120], rng.integers(0, 4, size=60).astype(np.uint8), t[120:]])
    sc, cg = O.ksw_extd2_global(q, t, MAT)
    assert sc == 300 - (41 + 60)
    assert [(int(c) >> 4, "MID"[c & 0xf]) for c in cg if (c & 0xf) == 1] == [(60, "I")]
    # N scores 0 against anything (caller.cpp:336-337)
    q = t.copy(); q[10] = 4
    assert O.ksw_extd2_global(q, t, MAT)[0] == 299


def test_gaps_are_left_aligned():
    # deleting one copy of a tandem repeat unit: ksw2 (no KSW_EZ_RIGHT) puts the gap leftmost
    unit = np.array([0, 1, 2], dtype=np.uint8)
    left = np.array([3, 3, 1, 0, 2, 3, 1, 2, 0, 0], dtype=np.uint8)
    right = np.array([1, 3, 3, 0, 2, 1, 0, 3, 2, 2], dtype=np.uint8)
    t = np.concatenate([left, np.tile(unit, 30), right])
    q = np.concatenate([left, np.tile(unit, 10), right])
    sc, cg = O.ksw_extd2_global(q, t, MAT)
    assert caller.cigar_string(cg) == "10M60D40M"
    t2, q2 = q, t          # and the insertion case
    sc, cg = O.ksw_extd2_global(q2, t2, MAT)
    assert caller.cigar_string(cg) == "10M60I40M"


def test_empty_inputs():
    t = np.array([0, 1, 2], dtype=np.uint8)
    assert O.ksw_extd2_global(np.zeros(0, np.uint8), t, MAT) == (0, pytest.approx(np.zeros(0)))


def test_fuzz_ratio_closed_form():
    assert O.fuzz_ratio(b"", b"") == 100.0
    assert O.fuzz_ratio(b"ACGT", b"") == 0.0
    assert O.fuzz_ratio(b"ACGT", b"ACGT") == 100.0
    assert O.lcs(b"AGGTAB", b"GXTXAYB") == 4
    # "this is a test" vs "this is a test!" is rapidfuzz's documented example: 96.55...
    r = O.fuzz_ratio(b"this is a test", b"this is a test!")
    assert abs(r - 96.55172413793103) < 1e-12
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = bytes(rng.integers(65, 69, size=int(rng.integers(0, 60))).astype(np.uint8))
        b = bytes(rng.integers(65, 69, size=int(rng.integers(0, 60))).astype(np.uint8))
        l = O.lcs(a, b)
        # brute-force LCS
        L = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
        for i in range(len(a)):
            for j in range(len(b)):
                L[i + 1][j + 1] = L[i][j] + 1 if a[i] == b[j] else max(L[i][j + 1], L[i + 1][j])
        assert l == L[len(a)][len(b)]
        tot = len(a) + len(b)
        want = (1.0 - ((tot - 2 * l) / tot if tot else 0.0)) * 100.0
        assert O.fuzz_ratio(a, b) == want
