"""BGZF blocks inflated on the GPU: Python binding of svdss_bgzf_inflate (csrc/inflate.hip) and the block walk
htslib's bgzf_read_block does (/root/reference/ping_pong.cpp:58,247-249 reach it through sam_read1)."""
import struct


def bgzf_blocks(data):
    """(coff, clen, isize, crc) of every BGZF block of `data` (bytes of a .bam / .gz written by bgzip): the raw deflate
    stream is data[coff:coff + clen].  The walk htslib's bgzf_read_block does (header, BC subfield, footer)."""
    out = []
    pos, n = 0, len(data)
    while pos + 18 <= n:
        if data[pos:pos + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("bad BGZF block at %d" % pos)
        xlen = struct.unpack_from("<H", data, pos + 10)[0]
        bsize, o = -1, 0
        while o + 4 <= xlen:
            si1, si2, slen = data[pos + 12 + o], data[pos + 13 + o], struct.unpack_from("<H", data, pos + 14 + o)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", data, pos + 16 + o)[0]
                break
            o += 4 + slen
        if bsize < 0:
            raise ValueError("BGZF block without BC field")
        clen = bsize + 1 - 12 - xlen - 8
        crc, isize = struct.unpack_from("<II", data, pos + 12 + xlen + clen)
        out.append((pos + 12 + xlen, clen, isize, crc))
        pos += bsize + 1
    return out


def gpu_inflate(data, blocks, device=0, uoff=None):
    """Inflates raw deflate streams on the GPU: blocks = [(coff, clen, isize)], block i goes to uoff[i] (default: back
    to back).  Returns the inflated bytes (numpy uint8).  Raises SvdssError (code SVDSS_EIO) with .bad_block set when a
    stream does not inflate to its isize."""
    import ctypes as C
    import numpy as np
    from ._lib import lib, SvdssError
    n = len(blocks)
    if uoff is None:
        uoff, acc = [], 0
        for b in blocks:
            uoff.append(acc)
            acc += b[2]
    total = max([u + b[2] for u, b in zip(uoff, blocks)], default=0)
    rec = np.zeros(n, dtype=[("coff", "<i8"), ("clen", "<i4"), ("isize", "<i4"), ("uoff", "<i8")])
    for i, b in enumerate(blocks):
        rec[i] = (b[0], b[1], b[2], uoff[i])
    comp = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(max(total, 1), dtype=np.uint8)
    d_out = C.c_void_p()
    rc = lib.svdss_device_alloc(device, max(total, 16), C.byref(d_out))
    if rc:
        raise SvdssError(rc, "svdss_device_alloc")
    obj = C.c_void_p()
    bad = C.c_int64(-1)
    lib.svdss_device_memset(device, d_out, 0, max(total, 16))
    try:
        rc = lib.svdss_bgzf_inflate(C.byref(obj), device, comp.ctypes.data, len(comp), rec.ctypes.data, n, d_out,
                                    out.ctypes.data, total, C.byref(bad))
        if rc:
            e = SvdssError(rc, "svdss_bgzf_inflate")
            e.bad_block = bad.value
            raise e
    finally:
        lib.svdss_inflate_free(obj)
        lib.svdss_device_free(device, d_out)
    return out[:total]


def gpu_deflate(data, block_bytes=0xff00, device=0, return_stats=False, dense=False):
    """BGZF members written by the GPU encoder (svdss_bgzf_deflate, csrc/deflate.hip) for `data` cut into blocks of
    block_bytes; the CRC32 / ISIZE footers are filled in here, as the binary's writer does (csrc/bam_writer.h).
    Returns the concatenated members (bytes) -- a valid BGZF stream without the EOF marker block.  dense: the library
    writes the members back to back itself (out_stride 0, what the binary's writer asks for) instead of one per stride."""
    import ctypes as C
    import zlib
    import numpy as np
    from ._lib import check, lib
    raw = np.frombuffer(bytes(data), dtype=np.uint8)
    n = len(raw)
    if n == 0:
        return (b"", {}) if return_stats else b""
    nb = (n + block_bytes - 1) // block_bytes
    stride = 0 if dense else 0x10000 + 64
    out = np.zeros(nb * (block_bytes + 128) if dense else nb * stride, dtype=np.uint8)
    lens = np.zeros(nb, dtype=np.int32)
    obj = C.c_void_p()
    try:
        check(lib.svdss_bgzf_deflate(C.byref(obj), device, raw.ctypes.data, n, block_bytes, out.ctypes.data, stride,
                                     lens.ctypes.data), "svdss_bgzf_deflate")
        ms = lib.svdss_deflate_kernel_ms(obj)
    finally:
        lib.svdss_deflate_free(obj)
    parts = []
    at = np.concatenate([[0], np.cumsum(lens)]) if dense else np.arange(nb + 1) * stride
    for i in range(nb):
        blk = raw[i * block_bytes:(i + 1) * block_bytes]
        m = bytearray(out[int(at[i]):int(at[i]) + int(lens[i])].tobytes())
        m[-8:-4] = struct.pack("<I", zlib.crc32(blk.tobytes()) & 0xffffffff)
        m[-4:] = struct.pack("<I", len(blk))
        parts.append(bytes(m))
    res = b"".join(parts)
    return (res, {"kernel_ms": ms, "members": nb}) if return_stats else res
