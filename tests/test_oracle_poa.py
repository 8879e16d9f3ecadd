"""POA consensus oracle (oracle/svdss_oracle_poa.c): invariants of the abPOA step at
/root/reference/caller.cpp:257-308 and the stated tolerance against the truth."""
import numpy as np
import pytest

from tests import oracle_lib as O


def mutate(rng, s, err):
    out = []
    for b in s.tolist():
        u = rng.random()
        if u < err * 0.4:
            out.append((b + int(rng.integers(1, 4))) % 4)
        elif u < err * 0.7:
            out += [b, int(rng.integers(0, 4))]
        elif u < err:
            pass
        else:
            out.append(b)
    return np.array(out, dtype=np.uint8)


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[len(b)]


def test_trivial_inputs():
    rng = np.random.default_rng(1)
    t = rng.integers(0, 4, size=300).astype(np.uint8)
    assert (O.poa_consensus([t]) == t).all()                        # one read: itself
    assert (O.poa_consensus([t.copy() for _ in range(5)]) == t).all()  # identical reads: that read
    assert len(O.poa_consensus([])) == 0
    n = t.copy(); n[10] = 4
    c = O.poa_consensus([n, t, t])
    assert (c == t).all()                                           # N scores 0, the majority base wins


def test_majority_rules():
    rng = np.random.default_rng(2)
    t = rng.integers(0, 4, size=500).astype(np.uint8)
    b = t.copy(); b[250] = (b[250] + 1) % 4
    assert (O.poa_consensus([t, b, b]) == b).all() and (O.poa_consensus([b, t, t]) == t).all()
    ins = np.concatenate([t[:200], rng.integers(0, 4, size=40).astype(np.uint8), t[200:]])
    assert (O.poa_consensus([t, ins, ins, ins, t]) == ins).all()   # insertion carried by most reads appears
    assert (O.poa_consensus([ins, t, t, t, ins]) == t).all()
    dl = np.concatenate([t[:250], t[330:]])                         # 80-bp deletion, far beyond the band w = 15
    assert (O.poa_consensus([t, dl, dl]) == dl).all()
    assert (O.poa_consensus([dl, t, t]) == t).all()


@pytest.mark.parametrize("n,err,length", [(5, 0.01, 800), (10, 0.02, 1500), (25, 0.05, 600), (3, 0.005, 2500)])
def test_consensus_within_tolerance_of_truth(n, err, length):
    rng = np.random.default_rng(n * 1000 + length)
    t = rng.integers(0, 4, size=length).astype(np.uint8)
    reads = [mutate(rng, t, err) for _ in range(n)]
    c = O.poa_consensus(reads)
    assert edit_distance(c.tolist(), t.tolist()) <= max(2, int(0.005 * length))


def test_order_dependence_is_deterministic():
    rng = np.random.default_rng(9)
    t = rng.integers(0, 4, size=400).astype(np.uint8)
    reads = [mutate(rng, t, 0.03) for _ in range(6)]
    a, b = O.poa_consensus(reads), O.poa_consensus(reads)
    assert (a == b).all()
