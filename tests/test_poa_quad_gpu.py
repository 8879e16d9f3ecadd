"""The several-sub-clusters-per-wavefront POA kernel (csrc/poa_quad.hip) on the GPU: its cross-lane primitives against the
definitions the CPU wave emulator uses (tests/native/wave_emu.h, tests/test_poa_quad_emu.py), and whole sub-clusters
against the oracle for every group width."""
import numpy as np
import pytest

from svdss_amd import _lib
from tests import oracle_lib as O
from tests.mirror import caller
from tests.test_oracle_poa import mutate
from tests.test_poa_quad_emu import _noisy

pytestmark = pytest.mark.gpu
LET = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _to_str(a):
    return bytes(LET[a]).decode()


def test_group_primitives_on_the_device():
    rng = np.random.default_rng(3)
    for trial in range(4):
        x = rng.integers(-1000, 1000, size=64).astype(np.int32)
        if trial == 1:
            x[17] = 12345
        out = np.zeros((3, 64, 12), dtype=np.int32)
        assert _lib.lib.svdss_poa_quad_selftest(x.ctypes.data, out.ctypes.data, 0) == 0
        for gi, gw in enumerate((16, 32, 64)):
            for lane in range(64):
                g, l = divmod(lane, gw)
                grp = x[g * gw:(g + 1) * gw]
                o = out[gi, lane]
                assert o[0] == (-7 if l == 0 else x[lane - 1]), (gw, lane, "shr1")
                assert o[1] == (-9 if l == gw - 1 else x[lane + 1]), (gw, lane, "shl1")
                assert o[2] == grp[:l + 1].max(), (gw, lane, "scan_max")
                assert o[3] == grp[:l + 1].sum(), (gw, lane, "scan_add")
                assert o[4] == grp.max() and o[5] == grp.min(), (gw, lane, "all_max / all_min")
                assert o[6] == grp[-1], (gw, lane, "last")
                assert o[7] == grp[(int(grp[-1]) >> 3) & (gw - 1)], (gw, lane, "from")
                bits = sum(1 << i for i in range(gw) if grp[i] & 1)
                assert (int(o[8]) & 0xffffffff) | ((int(o[9]) & 0xffffffff) << 32) == bits, (gw, lane, "bits")
                assert o[10] == (1 if (x == 12345).any() else 0)
                inv = ~(bits >> l) & ((1 << 64) - 1)
                assert o[11] == (inv & -inv).bit_length() - 1, (gw, lane, "ctz64")


@pytest.mark.parametrize("gw", [16, 32, 64])
def test_quad_stage_is_the_specification(monkeypatch, gw):
    monkeypatch.setenv("SVDSS_POA_QUAD_GW", str(gw))
    clusters = _noisy(300 + gw, 70, 60, 700)
    rng = np.random.default_rng(gw)
    for _ in range(12):                                    # bench-shaped: 15-30 reads of 0.6-2.6 kb, 0.5 % substitutions
        t = rng.integers(0, 4, size=int(rng.integers(600, 2600))).astype(np.uint8)
        clusters.append([mutate(rng, t, 0.005) for _ in range(int(rng.integers(15, 31)))])
    clusters += [[], [np.array([0, 1, 2, 3], np.uint8)], [np.zeros(0, np.uint8), np.array([1, 1], np.uint8)]]
    want = [_to_str(O.poa_consensus(cl)) for cl in clusters]
    got, stats = caller.run_poa(clusters)
    assert got == want
    assert stats["quad_back"] <= len(clusters) // 3, stats      # the first stage is the one that ran
    monkeypatch.setenv("SVDSS_POA_QUAD", "0")
    got0, stats0 = caller.run_poa(clusters)
    assert got0 == want and stats0["quad_back"] == 0
