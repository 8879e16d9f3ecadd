"""ctypes wrapper of tests/native/_poa_quad_emu.so: the device code of the several-sub-clusters-per-wavefront POA kernel
(svdss_amd/csrc/poa_quad_core.h) on the CPU wave emulator.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tests", "native", "_poa_quad_emu.so")
_SRC = [os.path.join(ROOT, "tests", "native", "poa_quad_emu.cpp"), os.path.join(ROOT, "tests", "native", "wave_emu.h"),
        os.path.join(ROOT, "tests", "native", "poa_quad_backend_emu.h"),
        os.path.join(ROOT, "svdss_amd", "csrc", "poa_quad_core.h"), os.path.join(ROOT, "svdss_amd", "csrc", "poa_quad_defs.h"),
        os.path.join(ROOT, "svdss_amd", "csrc", "poa_task.h")]


def _build():
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in _SRC):
        return
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", _SO, _SRC[0]])


_build()
_lib = C.CDLL(_SO)
_lib.poaq_emu_consensus.restype = C.c_int
_lib.poaq_emu_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]


def run(clusters, gw=16, c=4):
    """clusters: list of lists of uint8 arrays (symbols 0..4).  Returns (list of consensus arrays or None where the kernel
    handed the sub-cluster back, status array, cells)."""
    seqs = [np.asarray(r, dtype=np.uint8) for cl in clusters for r in cl]
    seq_off = np.zeros(len(seqs) + 1, dtype=np.int64)
    seq_off[1:] = np.cumsum([len(s) for s in seqs])
    flat = np.ascontiguousarray(np.concatenate(seqs)) if seqs and seq_off[-1] else np.zeros(1, np.uint8)
    cluster_off = np.zeros(len(clusters) + 1, dtype=np.int64)
    cluster_off[1:] = np.cumsum([len(cl) for cl in clusters])
    n = len(clusters)
    out_len = np.zeros(n, dtype=np.int64)
    st = np.zeros(n, dtype=np.int32)
    cap = int(seq_off[-1]) + 16
    cons = np.zeros(cap, dtype=np.uint8)
    cons_off = np.zeros(n, dtype=np.int64)
    cells = C.c_ulonglong(0)
    rc = _lib.poaq_emu_consensus(flat.ctypes.data, seq_off.ctypes.data, cluster_off.ctypes.data, n, gw, c, out_len.ctypes.data,
                                 st.ctypes.data, cons.ctypes.data, cons_off.ctypes.data, cap, C.byref(cells))
    if rc != 0:
        raise RuntimeError(f"poaq_emu_consensus: {rc}")
    res = [cons[cons_off[k]:cons_off[k] + out_len[k]].copy() if st[k] == 0 else None for k in range(n)]
    return res, st, cells.value
