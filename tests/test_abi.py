"""The C-ABI library loads, exports every symbol include/svdss_hip.h declares,
and fails loudly (no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import svdss_amd
from svdss_amd import _lib
from tests.common import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "svdss_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svdss_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} not bound in svdss_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_strerror_and_nt6():
    assert _lib.lib.svdss_strerror(0) == b"ok"
    assert b"HIP" in _lib.lib.svdss_strerror(4)
    assert svdss_amd.nt6_encode("ACGTNacgtnXY").tolist() == [1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 5, 5]
    assert (svdss_amd.NT6_TABLE[np.frombuffer(b"ACGTNacgtn\x00", np.uint8)] ==
            np.array([1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 0])).all()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    from svdss_amd import synth
    ref = synth.make_reference([5000], seed=1)
    ix = svdss_amd.FMDIndex.build(ref, threads=1)
    pp = svdss_amd.PingPong(ix)
    flat, offs = svdss_amd.pack_reads([ref[0][:200]])
    with pytest.raises(svdss_amd.SvdssError) as e:
        pp.ping_pong_search(flat, offs)          # index not on a device
    assert e.value.code == 5
    with pytest.raises(svdss_amd.SvdssError) as e:
        ix.to_device(0)                           # no device to put it on
    assert e.value.code in (2, 4)
