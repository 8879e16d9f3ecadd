"""BAM records walked, filtered and searched on the GPU: Python binding of svdss_bam_batch_run (csrc/bam_device.hip),
which stands where PingPong::load_batch_bam + process_batch stand (/root/reference/ping_pong.cpp:53-128,176-209).
The host side here does what the binary's scanner does: find the BGZF blocks, read the BAM header, cut the file into
batches of consecutive blocks."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import bgzf
from ._lib import SVDSS_BAM_PUTATIVE, SVDSS_SFS_ASSEMBLE, SvdssError, lib


class BamResult(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("n_slots", C.c_int64), ("n_searched", C.c_int64), ("n_short", C.c_int64),
                ("total_sfs", C.c_int64), ("name_off", C.POINTER(C.c_int32)), ("names", C.POINTER(C.c_char)),
                ("hp", C.POINTER(C.c_int32)), ("sidx", C.POINTER(C.c_int32)), ("counts", C.POINTER(C.c_int64)),
                ("qs", C.POINTER(C.c_int32)), ("len", C.POINTER(C.c_int32)), ("inflate_kernel_ms", C.c_double),
                ("stage_ms", C.c_double * 8)]


def bam_header(data, blocks):
    """(n_ref, header bytes of the inflated stream) -- inflates leading blocks on the host until the header is complete."""
    buf = b""
    k = 0

    def need(n):
        nonlocal buf, k
        while len(buf) < n:
            if k >= len(blocks):
                raise ValueError("truncated header")
            coff, clen, isize, _ = blocks[k]
            buf += zlib.decompress(bytes(data[coff:coff + clen]), -15)
            k += 1
    need(12)
    if buf[:4] != b"BAM\1":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    need(12 + l_text)
    n_ref = struct.unpack_from("<i", buf, 8 + l_text)[0]
    o = 12 + l_text
    for _ in range(n_ref):
        need(o + 4)
        l_name = struct.unpack_from("<i", buf, o)[0]
        o += 4 + l_name + 4
        need(o)
    return n_ref, o


def search_bam(index, data, assemble=True, putative=False, batch_bytes=256 << 20, n_ref=None, skip=None):
    """`data`: the bytes of a BAM file.  Runs the whole file through svdss_bam_batch_run in batches of about
    batch_bytes inflated bytes (one after the other; the binary runs several at once) and returns a list of
    (name, hp, None | [(qs, len), ...]) for every record that passed the filters, in file order, plus a dict of counters."""
    blocks = bgzf.bgzf_blocks(data)
    if n_ref is None or skip is None:
        n_ref, skip = bam_header(data, blocks)
    comp = np.frombuffer(bytes(data), dtype=np.uint8)
    stream = C.c_void_p()
    rc = lib.svdss_bam_stream_create(n_ref, C.byref(stream))
    if rc:
        raise SvdssError(rc, "svdss_bam_stream_create")
    batch = C.c_void_p()
    out, stats = [], {"records": 0, "short": 0, "batches": 0}
    flags = (SVDSS_SFS_ASSEMBLE if assemble else 0) | (SVDSS_BAM_PUTATIVE if putative else 0)
    try:
        groups, cur, acc = [], [], 0
        for b in blocks:
            cur.append(b)
            acc += b[2]
            if acc >= batch_bytes:
                groups.append(cur)
                cur, acc = [], 0
        groups.append(cur)   # (possibly empty: the last batch closes the stream)
        for seq, g in enumerate(groups):
            rec = np.zeros(max(1, len(g)), dtype=[("coff", "<i8"), ("clen", "<i4"), ("isize", "<i4"), ("uoff", "<i8")])
            crc = np.zeros(max(1, len(g)), dtype=np.uint32)
            for i, b in enumerate(g):
                rec[i] = (b[0], b[1], b[2], 0)
                crc[i] = b[3]
            comp_p = (C.c_void_p * 1)(comp.ctypes.data)
            comp_n = (C.c_int64 * 1)(len(comp))
            blk_p = (C.c_void_p * 1)(rec.ctypes.data)
            crc_p = (C.c_void_p * 1)(crc.ctypes.data)
            nb = (C.c_int64 * 1)(len(g))
            rc = lib.svdss_bam_batch_run(stream, seq, 1 if seq == len(groups) - 1 else 0, skip if seq == 0 else 0, index._h, 1,
                                         comp_p, comp_n, blk_p, crc_p, nb, flags, C.byref(batch))
            if rc:
                e = SvdssError(rc, "svdss_bam_batch_run")
                e.detail = (lib.svdss_bam_batch_error(batch) or b"").decode() if batch else ""
                if not e.detail:
                    e.detail = lib.svdss_bam_stream_error(stream).decode()
                raise e
            r = BamResult()
            lib.svdss_bam_batch_result(batch, C.byref(r))
            stats["records"] += r.n_records
            stats["short"] += r.n_short
            stats["batches"] += 1
            n = r.n_slots
            name_off = np.ctypeslib.as_array(r.name_off, shape=(n + 1,)).copy() if n else np.zeros(1, np.int32)
            names = C.string_at(r.names, int(name_off[-1])) if n else b""
            hp = np.ctypeslib.as_array(r.hp, shape=(n,)).copy() if n else np.zeros(0, np.int32)
            sidx = np.ctypeslib.as_array(r.sidx, shape=(n,)).copy() if n else np.zeros(0, np.int32)
            counts = np.ctypeslib.as_array(r.counts, shape=(r.n_searched,)).copy() if r.n_searched else np.zeros(0, np.int64)
            qs = np.ctypeslib.as_array(r.qs, shape=(r.total_sfs,)).copy() if r.total_sfs else np.zeros(0, np.int32)
            ln = np.ctypeslib.as_array(r.len, shape=(r.total_sfs,)).copy() if r.total_sfs else np.zeros(0, np.int32)
            first = np.concatenate([[0], np.cumsum(counts)])
            for i in range(n):
                nm = names[name_off[i]:name_off[i + 1]].decode()
                k = int(sidx[i])
                sfs = None if k < 0 else [(int(qs[j]), int(ln[j])) for j in range(int(first[k]), int(first[k + 1]))]
                out.append((nm, int(hp[i]), sfs))
        nseg = C.c_int64(0)
        stats["rewalked"] = lib.svdss_bam_stream_rewalked(stream, C.byref(nseg))
        stats["segments"] = nseg.value
    finally:
        if batch:
            lib.svdss_bam_batch_free(batch)
        lib.svdss_bam_stream_free(stream)
    return out, stats


class BamSelection(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("n_selected", C.c_int64), ("n_bytes", C.c_int64), ("rec_off", C.POINTER(C.c_int64)),
                ("bytes", C.POINTER(C.c_uint8)), ("inflate_kernel_ms", C.c_double), ("stage_ms", C.c_double * 8),
                ("slim", C.c_int32)]


def select_bam(data, names=None, regions=None, min_mapq=0, batch_bytes=256 << 20, device=0):
    """svdss_bam_select_run over a whole BAM file (bytes): the records `SVDSS call` keeps -- no flag 4 / 256 / 2048,
    mapq >= min_mapq, and (when given) read name in `names` or alignment overlapping one of `regions` =
    [(tid, beg, end)] sorted by (tid, beg).  Returns ([record bytes without the block_size field], counters)."""
    blocks = bgzf.bgzf_blocks(data)
    n_ref, skip = bam_header(data, blocks)
    comp = np.frombuffer(bytes(data), dtype=np.uint8)
    nm = b"".join(n.encode() if isinstance(n, str) else n for n in (names or []))
    nm_off = np.zeros(len(names or []) + 1, dtype=np.int64)
    if names:
        nm_off[1:] = np.cumsum([len(n) for n in names])
    rt = np.array([r[0] for r in (regions or [])], dtype=np.int32)
    rb = np.array([r[1] for r in (regions or [])], dtype=np.int32)
    re_ = np.array([r[2] for r in (regions or [])], dtype=np.int32)
    flt = C.c_void_p()
    rc = lib.svdss_bam_filter_create(device, min_mapq, n_ref, nm if names else None, nm_off.ctypes.data if names else None, len(names or []),
                                     rt.ctypes.data if regions else None, rb.ctypes.data if regions else None,
                                     re_.ctypes.data if regions else None, len(regions or []), C.byref(flt))
    if rc:
        raise SvdssError(rc, "svdss_bam_filter_create")
    stream = C.c_void_p()
    lib.svdss_bam_stream_create(n_ref, C.byref(stream))
    batch = C.c_void_p()
    out, stats = [], {"records": 0, "batches": 0}
    try:
        groups, cur, acc = [], [], 0
        for b in blocks:
            cur.append(b)
            acc += b[2]
            if acc >= batch_bytes:
                groups.append(cur)
                cur, acc = [], 0
        groups.append(cur)
        for seq, g in enumerate(groups):
            rec = np.zeros(max(1, len(g)), dtype=[("coff", "<i8"), ("clen", "<i4"), ("isize", "<i4"), ("uoff", "<i8")])
            crc = np.zeros(max(1, len(g)), dtype=np.uint32)
            for i, b in enumerate(g):
                rec[i] = (b[0], b[1], b[2], 0)
                crc[i] = b[3]
            rc = lib.svdss_bam_select_run(stream, seq, 1 if seq == len(groups) - 1 else 0, skip if seq == 0 else 0, flt, 1,
                                          (C.c_void_p * 1)(comp.ctypes.data), (C.c_int64 * 1)(len(comp)), (C.c_void_p * 1)(rec.ctypes.data),
                                          (C.c_void_p * 1)(crc.ctypes.data), (C.c_int64 * 1)(len(g)), C.byref(batch))
            if rc:
                e = SvdssError(rc, "svdss_bam_select_run")
                e.detail = (lib.svdss_bam_batch_error(batch) or b"").decode() if batch else ""
                raise e
            r = BamSelection()
            lib.svdss_bam_batch_selection(batch, C.byref(r))
            stats["records"] += r.n_records
            stats["batches"] += 1
            if r.n_selected:
                off = np.ctypeslib.as_array(r.rec_off, shape=(r.n_selected + 1,))
                raw = C.string_at(r.bytes, r.n_bytes)
                for k in range(r.n_selected):
                    o = int(off[k])
                    bs = struct.unpack_from("<i", raw, o)[0]
                    out.append(raw[o + 4:o + 4 + bs])
    finally:
        if batch:
            lib.svdss_bam_batch_free(batch)
        lib.svdss_bam_stream_free(stream)
        lib.svdss_bam_filter_free(flt)
    return out, stats


def select_bam_store(data, names, regions, min_mapq=0, batch_bytes=256 << 20, device=0, max_store_bytes=1 << 34):
    """`SVDSS call`'s ONE pass: svdss_bam_select_store_run over a whole BAM file with `names` as the filter (the first pass'
    records come back whole) and every record that passes the flag / mapq filters kept, slim, in a svdss_bam_store_t; then
    svdss_bam_store_select of every stored batch with `regions` [(tid, beg, end)] sorted by (tid, beg).
    Returns (named records, slim records that overlap a region -- both without the block_size field --, counters)."""
    blocks = bgzf.bgzf_blocks(data)
    n_ref, skip = bam_header(data, blocks)
    comp = np.frombuffer(bytes(data), dtype=np.uint8)
    nm = b"".join(n.encode() if isinstance(n, str) else n for n in names)
    nm_off = np.zeros(len(names) + 1, dtype=np.int64)
    nm_off[1:] = np.cumsum([len(n) for n in names])
    rt = np.array([r[0] for r in regions], dtype=np.int32)
    rb = np.array([r[1] for r in regions], dtype=np.int32)
    re_ = np.array([r[2] for r in regions], dtype=np.int32)
    f_names, f_regions, store = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.svdss_bam_filter_create(device, min_mapq, n_ref, nm, nm_off.ctypes.data, len(names), None, None, None, 0, C.byref(f_names))
    if rc:
        raise SvdssError(rc, "svdss_bam_filter_create")
    rc = lib.svdss_bam_filter_create(device, min_mapq, n_ref, None, None, 0, rt.ctypes.data if regions else None, rb.ctypes.data if regions else None,
                                     re_.ctypes.data if regions else None, len(regions), C.byref(f_regions))
    if rc:
        raise SvdssError(rc, "svdss_bam_filter_create")
    rc = lib.svdss_bam_store_create(device, max_store_bytes, min(max_store_bytes, 1 << 20), C.byref(store))
    if rc:
        raise SvdssError(rc, "svdss_bam_store_create")
    stream = C.c_void_p()
    lib.svdss_bam_stream_create(n_ref, C.byref(stream))
    batch = C.c_void_p()
    named, slim, stats = [], [], {"records": 0, "batches": 0}

    def take(r, into):
        if r.n_selected:
            off = np.ctypeslib.as_array(r.rec_off, shape=(r.n_selected + 1,))
            raw = C.string_at(r.bytes, r.n_bytes)
            for k in range(r.n_selected):
                o = int(off[k])
                bs = struct.unpack_from("<i", raw, o)[0]
                into.append(raw[o + 4:o + 4 + bs])
    try:
        groups, cur, acc = [], [], 0
        for b in blocks:
            cur.append(b)
            acc += b[2]
            if acc >= batch_bytes:
                groups.append(cur)
                cur, acc = [], 0
        groups.append(cur)
        for seq, g in enumerate(groups):
            rec = np.zeros(max(1, len(g)), dtype=[("coff", "<i8"), ("clen", "<i4"), ("isize", "<i4"), ("uoff", "<i8")])
            crc = np.zeros(max(1, len(g)), dtype=np.uint32)
            for i, b in enumerate(g):
                rec[i] = (b[0], b[1], b[2], 0)
                crc[i] = b[3]
            rc = lib.svdss_bam_select_store_run(stream, seq, 1 if seq == len(groups) - 1 else 0, skip if seq == 0 else 0, f_names, store, 1,
                                                (C.c_void_p * 1)(comp.ctypes.data), (C.c_int64 * 1)(len(comp)), (C.c_void_p * 1)(rec.ctypes.data),
                                                (C.c_void_p * 1)(crc.ctypes.data), (C.c_int64 * 1)(len(g)), C.byref(batch))
            if rc:
                e = SvdssError(rc, "svdss_bam_select_store_run")
                e.detail = (lib.svdss_bam_batch_error(batch) or b"").decode() if batch else ""
                raise e
            r = BamSelection()
            lib.svdss_bam_batch_selection(batch, C.byref(r))
            # (with a store the first pass' records are slim too; once it is over its limit the batches behind come whole)
            stats["records"] += r.n_records
            stats["batches"] += 1
            stats.setdefault("named_slim", []).extend([int(r.slim)] * int(r.n_selected))
            take(r, named)
        complete, n_rec, n_bytes = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        n_b = lib.svdss_bam_store_batches(store, C.byref(complete), C.byref(n_rec), C.byref(n_bytes))
        stats.update({"stored_batches": n_b, "complete": complete.value, "stored_records": n_rec.value, "stored_bytes": n_bytes.value})
        if complete.value:
            for seq in range(n_b):
                rc = lib.svdss_bam_store_select(store, seq, f_regions, C.byref(batch))
                if rc:
                    raise SvdssError(rc, "svdss_bam_store_select")
                r = BamSelection()
                lib.svdss_bam_batch_selection(batch, C.byref(r))
                assert r.slim == 1
                take(r, slim)
    finally:
        if batch:
            lib.svdss_bam_batch_free(batch)
        lib.svdss_bam_stream_free(stream)
        lib.svdss_bam_filter_free(f_names)
        lib.svdss_bam_filter_free(f_regions)
        lib.svdss_bam_store_free(store)
    return named, slim, stats
