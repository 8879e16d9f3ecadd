"""Host-side mirror of the reference's `search` stage over the C-ABI.

Names follow /root/reference: `FMDIndex` stands for the rb3_fmi_t restored at
ping_pong.cpp:245, `PingPong.ping_pong_search` for ping_pong.cpp:4-49 (batched),
`PingPong.process_batch` for ping_pong.cpp:176-209 (XF/HP handling),
`output_batch` for ping_pong.cpp:213-236 and `parse_sfsfile` for sfs.cpp:5-30.
All compute goes through libsvdss_hip.so; nothing here falls back to the CPU.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import SVDSS_SFS_ASSEMBLE, SvdssError, check, lib

# ping_pong.hpp:46-52 as a 256-entry table (bytes >= 128 cannot come out of
# seq_nt16_str / kseq; they map to 5 like every other non-ACGT byte)
NT6_TABLE = np.full(256, 5, dtype=np.uint8)
NT6_TABLE[0] = 0
for _ch, _v in (("A", 1), ("C", 2), ("G", 3), ("T", 4)):
    NT6_TABLE[ord(_ch)] = _v
    NT6_TABLE[ord(_ch.lower())] = _v


def nt6_encode(seq) -> np.ndarray:
    """a1: base -> nt6 through the library (svdss_nt6_encode)."""
    if isinstance(seq, str):
        seq = seq.encode()
    raw = bytes(seq)
    out = np.empty(len(raw), dtype=np.uint8)
    check(lib.svdss_nt6_encode(raw, len(raw), out.ctypes.data), "svdss_nt6_encode")
    return out


def _as_nt6(x) -> np.ndarray:
    if isinstance(x, (str, bytes, bytearray)):
        return nt6_encode(x)
    a = np.ascontiguousarray(x, dtype=np.uint8)
    return a


def pack_reads(reads: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate reads (str/bytes -> encoded, uint8 arrays taken as nt6) + offsets[n+1]."""
    enc = [_as_nt6(r) for r in reads]
    offsets = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc])
        flat = np.concatenate(enc) if offsets[-1] > 0 else np.zeros(0, dtype=np.uint8)
    else:
        flat = np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(flat, dtype=np.uint8), offsets


class FMDIndex:
    """FM-index over every record and its reverse complement (rb3_fmi_t role)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def build(cls, contigs: Sequence, threads: int = 0, device: Optional[int] = None) -> "FMDIndex":
        """`SVDSS index` (main.cpp:34-37).  device=None: host-side index (built on the GPU -- without one: SvdssError unless SVDSS_INDEX_CPU=1 --,
        then fetched); device=d: built in the HBM of GPU d and left resident there, k-mer table included."""
        enc = [_as_nt6(c) for c in contigs]
        lens = np.array([len(e) for e in enc], dtype=np.int64)
        flat = np.ascontiguousarray(np.concatenate(enc) if enc else np.zeros(0, np.uint8))
        if threads <= 0:
            import os
            threads = os.cpu_count() or 1
        h = C.c_void_p()
        if device is not None:
            check(lib.svdss_index_build_device(flat.ctypes.data, lens.ctypes.data, len(enc), threads, device,
                                               C.byref(h)), "svdss_index_build_device")
            return cls(h.value)
        check(lib.svdss_index_build(flat.ctypes.data, lens.ctypes.data, len(enc), threads, C.byref(h)),
              "svdss_index_build")
        return cls(h.value)

    @classmethod
    def load(cls, path: str) -> "FMDIndex":
        h = C.c_void_p()
        check(lib.svdss_index_load(path.encode(), C.byref(h)), "svdss_index_load")
        return cls(h.value)

    def save(self, path: str) -> None:
        check(lib.svdss_index_save(self._h, path.encode()), "svdss_index_save")

    def save_records(self, path: str) -> None:
        """The records file `SVDSS index` leaves beside the .fmd: load() of it rebuilds the index where it is made
        resident instead of reading text + suffix array from disk."""
        check(lib.svdss_index_save_records(self._h, path.encode()), "svdss_index_save_records")

    def save_fmd(self, path: str) -> None:
        """ropebwt3's rld0 dump (`ropebwt3 build -d` / upstream `SVDSS index`)."""
        check(lib.svdss_index_save_fmd(self._h, path.encode()), "svdss_index_save_fmd")

    def close(self) -> None:
        if self._h:
            lib.svdss_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        return lib.svdss_index_size(self._h)

    @property
    def acc(self) -> np.ndarray:
        a = np.zeros(7, dtype=np.int64)
        check(lib.svdss_index_acc(self._h, a.ctypes.data), "svdss_index_acc")
        return a

    @property
    def device_bytes(self) -> int:
        return lib.svdss_index_device_bytes(self._h)

    @property
    def kmer_k(self) -> int:
        return lib.svdss_index_kmer(self._h)

    @property
    def deep_frac(self) -> float:
        return lib.svdss_index_deep_frac(self._h)

    def bwt(self) -> np.ndarray:
        b = np.empty(self.size, dtype=np.uint8)
        check(lib.svdss_index_bwt(self._h, b.ctypes.data), "svdss_index_bwt")
        return b

    def count(self, pattern) -> int:
        p = _as_nt6(pattern)
        return lib.svdss_index_count(self._h, p.ctypes.data, len(p))

    def to_device(self, device: int = 0) -> "FMDIndex":
        check(lib.svdss_index_to_device(self._h, device), "svdss_index_to_device")
        return self

    def verify(self, stride: int = 1) -> dict:
        """The resident index against its own text, by direct comparison (svdss_index_verify_device): suffix-array
        rows strictly increasing as strings, BWT[i] == text[SA[i]-1], rank blocks, '$' rows, symbol histogram."""
        o = np.zeros(8, dtype=np.int64)
        check(lib.svdss_index_verify_device(self._h, stride, o.ctypes.data), "svdss_index_verify_device")
        keys = ("rows", "bad_order", "bad_bwt", "bad_range", "bad_block", "bad_dollar", "first_bad", "max_lcp")
        return dict(zip(keys, (int(x) for x in o)))


@dataclass
class SFSBatch:
    """Per-read SFS lists of one batch (solutions of process_batch)."""
    counts: np.ndarray   # int64[n_reads]
    qs: np.ndarray       # int32[total]
    len: np.ndarray      # int32[total]
    n_ext: np.ndarray    # int64[n_reads]
    kernel_ms: float = 0.0

    def per_read(self) -> List[List[Tuple[int, int]]]:
        out, o = [], 0
        for c in self.counts.tolist():
            out.append(list(zip(self.qs[o:o + c].tolist(), self.len[o:o + c].tolist())))
            o += c
        return out


class PingPong:
    """`SVDSS search` host logic on top of the HIP kernels."""

    def __init__(self, index: FMDIndex, assemble: bool = True, putative: bool = True):
        self.index = index
        self.assemble = assemble      # config.hpp:85, --noassemble
        self.putative = putative      # config.hpp:86, --noputative
        self._batch = C.c_void_p()

    def close(self):
        if self._batch:
            lib.svdss_sfs_batch_free(self._batch)
            self._batch = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fetch(self) -> SFSBatch:
        b = self._batch
        n = lib.svdss_sfs_batch_nreads(b)
        total = lib.svdss_sfs_batch_total(b)
        counts = np.zeros(n, dtype=np.int64)
        n_ext = np.zeros(n, dtype=np.int64)
        qs = np.zeros(total, dtype=np.int32)
        ln = np.zeros(total, dtype=np.int32)
        check(lib.svdss_sfs_batch_fetch(b, counts.ctypes.data, qs.ctypes.data, ln.ctypes.data,
                                        n_ext.ctypes.data), "svdss_sfs_batch_fetch")
        return SFSBatch(counts, qs, ln, n_ext, lib.svdss_sfs_batch_kernel_ms(b))

    def ping_pong_search(self, reads_nt6: np.ndarray, offsets: np.ndarray,
                         assemble: Optional[bool] = None) -> SFSBatch:
        """ping_pong.cpp:4-49 for every read of the batch (host buffers in, host arrays out)."""
        reads_nt6 = np.ascontiguousarray(reads_nt6, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        flags = SVDSS_SFS_ASSEMBLE if (self.assemble if assemble is None else assemble) else 0
        check(lib.svdss_sfs_search_batch(self.index._h, reads_nt6.ctypes.data, offsets.ctypes.data,
                                         len(offsets) - 1, flags, C.byref(self._batch)),
              "svdss_sfs_search_batch")
        return self._fetch()

    def ping_pong_search_device(self, d_reads_ptr: int, d_offsets_ptr: int, n_reads: int,
                                total_syms: int, stream: int = 0,
                                assemble: Optional[bool] = None, fetch: bool = True):
        """Same on buffers already resident in HBM (pointers as ints)."""
        flags = SVDSS_SFS_ASSEMBLE if (self.assemble if assemble is None else assemble) else 0
        check(lib.svdss_sfs_search_batch_device(self.index._h, C.c_void_p(d_reads_ptr),
                                                C.c_void_p(d_offsets_ptr), n_reads, total_syms, flags,
                                                C.c_void_p(stream), C.byref(self._batch)),
              "svdss_sfs_search_batch_device")
        return self._fetch() if fetch else None

    def device_results(self):
        """Last results as torch tensors aliasing the library's HBM buffers (counts, qs, len)."""
        import torch
        pc, pq, pl, pe = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.svdss_sfs_batch_device_ptrs(self._batch, C.byref(pc), C.byref(pq), C.byref(pl),
                                              C.byref(pe)), "svdss_sfs_batch_device_ptrs")
        n, total = lib.svdss_sfs_batch_nreads(self._batch), lib.svdss_sfs_batch_total(self._batch)

        class _Alias:
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": (shape,), "typestr": typestr,
                                                 "data": (ptr or 0, False), "version": 2}

        def wrap(ptr, count, typestr, dtype):
            if count == 0 or not ptr.value:
                return torch.zeros(0, dtype=dtype, device="cuda")
            return torch.as_tensor(_Alias(ptr.value, count, typestr), device="cuda")

        return (wrap(pc, n, "<i8", torch.int64), wrap(pq, total, "<i4", torch.int32),
                wrap(pl, total, "<i4", torch.int32))

    @property
    def last_total(self) -> int:
        return lib.svdss_sfs_batch_total(self._batch)

    @property
    def last_total_ext(self) -> int:
        return lib.svdss_sfs_batch_total_ext(self._batch)

    @property
    def last_segments(self) -> int:
        return lib.svdss_sfs_batch_segments(self._batch)

    @property
    def last_fallbacks(self) -> int:
        return lib.svdss_sfs_batch_fallbacks(self._batch)

    @property
    def last_used_bs(self) -> bool:
        return lib.svdss_sfs_batch_used_bs(self._batch) == 1

    @property
    def last_kernel_ms(self) -> float:
        return lib.svdss_sfs_batch_kernel_ms(self._batch)

    @property
    def last_search_kernel_ms(self) -> float:
        """HIP-event time of the search kernel alone (no ordering / stitching / assembling / gather)."""
        return lib.svdss_sfs_batch_search_kernel_ms(self._batch)

    def process_batch(self, names: Sequence[str], reads: Sequence, xf: Optional[Sequence[int]] = None,
                      hp: Optional[Sequence[int]] = None):
        """ping_pong.cpp:176-209: putative filter on XF (:202-203), HP carried as htag.

        Returns a list of (qname, htag, [(qs, l), ...]) for searched reads, in input order.
        """
        n = len(names)
        xf = [0] * n if xf is None else list(xf)   # XF missing => 0 (:196-198)
        hp = [0] * n if hp is None else list(hp)   # HP missing => 0 (:199-201)
        keep = [i for i in range(n) if not (self.putative and xf[i] != 0)]
        flat, offsets = pack_reads([reads[i] for i in keep])
        res = self.ping_pong_search(flat, offsets).per_read()
        return [(names[i], hp[i], res[j]) for j, i in enumerate(keep)]


def output_batch(solutions: Iterable[Tuple[str, int, List[Tuple[int, int]]]]) -> str:
    """ping_pong.cpp:224-230: `name|*\\tqs\\tl\\thtag\\t\\n`, '*' for repeated read name."""
    lines = []
    for qname, htag, sfs in solutions:
        first = True
        for qs, l in sfs:
            lines.append(f"{qname if first else '*'}\t{qs}\t{l}\t{htag}\t\n")
            first = False
    return "".join(lines)


def parse_sfsfile(text: str):
    """sfs.cpp:5-30: 4 whitespace-separated fields; '*' repeats the previous read name."""
    out, name = {}, None
    for line in text.splitlines():
        f = line.split()
        if len(f) < 4:
            continue
        if f[0] != "*":
            name = f[0]
            out[name] = []
        out[name].append((int(f[1]), int(f[2]), int(f[3])))
    return out
