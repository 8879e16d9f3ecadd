"""Builds and wraps tests/lane_emulator.cpp (CPU run of the kernel's per-lane code)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SVDSS_EMU_SANITIZE=1: the same code built with AddressSanitizer + UndefinedBehaviorSanitizer (the process must have
# libasan preloaded: tests/test_lane_logic.py::test_lane_code_under_sanitizers runs the module's tests that way)
SANITIZE = os.environ.get("SVDSS_EMU_SANITIZE") == "1"
SO = os.path.join(HERE, "_lane_emulator_san.so" if SANITIZE else "_lane_emulator.so")
SRC = os.path.join(HERE, "lane_emulator.cpp")
_deps = [SRC] + [os.path.join(HERE, "..", "svdss_amd", "csrc", f)
                 for f in ("sfs_core2.h", "sym_window.h", "fmd_layout.h", "index_host.h")]
if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in _deps):
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas"] +
                          (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if SANITIZE else ["-O2"]) +
                          ["-o", SO, SRC])
_lib = C.CDLL(SO)
_p, _i64 = C.c_void_p, C.c_int64
_lib.emu_search2.restype = _i64
_lib.emu_search2.argtypes = [_p, _p, _p, _i64, _i64, C.c_int, C.c_int, C.c_int, _p, _p, _p, _i64, _p, _p, C.c_int, _p]
_lib.emu_table.restype = C.c_int
_lib.emu_table.argtypes = [_p, C.c_int, _p, _p]


def kmer_table(index, K: int):
    """(lo, info) arrays of the 4^K entries the kernel's table builder (sv_table_entry) produces."""
    lo = np.zeros(1 << (2 * K), dtype=np.uint64)
    info = np.zeros(1 << (2 * K), dtype=np.uint64)
    assert _lib.emu_table(index._h, K, lo.ctypes.data, info.ctypes.data) == 0
    return lo, info


OP_NAMES = ["DONE", "LF", "TABLE", "SA", "TEXT", "FILL", "TEXT_SLOW", "PEEK", "SA_SET", "SET", "BS_SA", "BS_TEXT", "BS_TEXT_SLOW", "BS_ORD"]


def search2(index, flat: np.ndarray, offsets: np.ndarray, assemble: bool, K: int = 6, use_text: bool = True,
            n_seg: int = 1, use_set: bool = True, use_bs: bool = True):
    """v2 lane code (k-mer table of order K, LF, TEXT).  Returns (counts, qs, len, n_ext, op_counts)."""
    n = len(offsets) - 1
    total_syms = int(offsets[-1])
    padded = np.zeros(max(((total_syms + 15) // 16) * 16 + 16, 80), dtype=np.uint8)
    padded[:total_syms] = flat
    alloc_syms = len(padded) - 16
    cap = total_syms + n + 1
    counts = np.zeros(n, dtype=np.int64)
    n_ext = np.zeros(n, dtype=np.int64)
    ops = np.zeros(16, dtype=np.int64)
    seg_stats = np.zeros(2, dtype=np.int64)
    qs = np.zeros(cap, dtype=np.int32)
    ln = np.zeros(cap, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    t = _lib.emu_search2(index._h, padded.ctypes.data, offsets.ctypes.data, n, alloc_syms, int(assemble), K,
                         int(use_text) | (2 if (use_set and use_text) else 0) | (4 if (use_bs and use_text) else 0), counts.ctypes.data, qs.ctypes.data, ln.ctypes.data, cap,
                         n_ext.ctypes.data, ops.ctypes.data, n_seg, seg_stats.ctypes.data)
    assert t >= 0
    d = dict(zip(OP_NAMES, ops.tolist()))
    d["stitched"], d["fallback"] = seg_stats.tolist()
    return counts, qs[:t].copy(), ln[:t].copy(), n_ext, d
