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


def test_null_arguments_are_refused_not_dereferenced():
    """Every entry point of include/svdss_hip.h called with null pointers and zero counts (in a child process: a crash
    is the failure): status codes are errors or the empty-batch success, counters answer -1, frees are no-ops."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
from svdss_amd import _lib
ok_zero = {"svdss_indel_ratio_batch", "svdss_nt6_encode", "svdss_stream_destroy", "svdss_device_count", "svdss_bam_park_group_ready"}
for n in sorted(_lib.SIGNATURES):
    restype, argtypes = _lib.SIGNATURES[n]
    args = [a(0) if a in (C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_double) else None for a in argtypes]
    r = getattr(_lib.lib, n)(*args)
    if restype in (C.c_int, C.c_int32) and n not in ok_zero and not n.endswith(("_nreads", "_segments", "_hbm", "_kmer", "_nclusters", "_npairs", "_fallbacks")):
        assert r != 0, (n, r)
    if restype is C.c_int64 and n not in ok_zero:
        assert r == -1, (n, r)
print("all", len(_lib.SIGNATURES))
'''
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "all" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
