"""Minimal BAM/BGZF writer for tests (there is no samtools/htslib in the image)."""
import struct
import zlib

SEQ16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def bgzf(data: bytes, block=60000) -> bytes:
    out = [_bgzf_block(data[i:i + block]) for i in range(0, len(data), block)]
    out.append(_bgzf_block(b""))  # EOF marker
    return b"".join(out)


def record(qname: str, flag: int, tid: int, pos: int, mapq: int, cigar, seq: str, tags=(), qual: bytes = None) -> bytes:
    """cigar: list of (op_char, len); tags: list of (tag, type_char, value) with integer types cCsSiI or Z."""
    name = qname.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(op)) for op, l in cigar)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= SEQ16.get(ch.upper(), 15) << (4 if i % 2 == 0 else 0)
    qual = b"\xff" * l_seq if qual is None else qual
    aux = b""
    for tag, ty, val in tags:
        aux += tag.encode() + ty.encode()
        if ty == "Z":
            aux += val.encode() + b"\0"
        else:
            aux += struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[ty], val)
    core = struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, 4680, len(cigar), flag, l_seq, -1, -1, 0)
    body = core + name + cig + bytes(packed) + qual + aux
    return struct.pack("<i", len(body)) + body


def bam(refs, records) -> bytes:
    """refs: list of (name, length); records: list of bytes from record()."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    return bgzf(hdr + b"".join(records))


def reg2bin(beg: int, end: int) -> int:
    """SAM specification, section 5.3 (0-based half-open [beg, end))"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def reg2bin_csi(beg: int, end: int, min_shift: int, depth: int) -> int:
    """SAM specification, section 5.3, the general scheme of CSI (0-based half-open [beg, end))"""
    end -= 1
    s, t = min_shift, ((1 << (depth * 3)) - 1) // 7
    for l in range(depth, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << ((l - 1) * 3)
    return 0


def csi(bam_bytes: bytes, min_shift: int = 14, depth: int = 5) -> bytes:
    """A CSI index (CSIv1) of a coordinate-sorted BAM: the bins of the (min_shift, depth) scheme with their chunks and,
    per bin, loffset = the smallest virtual offset of a record that reaches into the bin's range; BGZF-compressed as
    `samtools index -c` writes it."""
    import zlib
    blocks, data, pos = [], bytearray(), 0
    while pos + 18 <= len(bam_bytes):
        bsize = struct.unpack_from("<H", bam_bytes, pos + 16)[0] + 1
        raw = zlib.decompress(bam_bytes[pos + 18:pos + bsize - 8], -15)
        blocks.append((pos, len(data), len(raw)))
        data += raw
        pos += bsize
    import bisect
    nonempty = [b for b in blocks if b[2] > 0]
    starts = [b[1] for b in nonempty]

    def voff(u):
        k = max(bisect.bisect_left(starts, u) - 1, 0)
        c, s, n = nonempty[k]
        return (c << 16) | (u - s)

    l_text = struct.unpack_from("<i", data, 4)[0]
    n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]
    p = 12 + l_text
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        p += 8 + l_name
    bins = [dict() for _ in range(n_ref)]
    loff = [dict() for _ in range(n_ref)]
    while p + 4 <= len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        tid, start = struct.unpack_from("<ii", data, p + 4)
        l_name = data[p + 12]
        n_cig = struct.unpack_from("<H", data, p + 16)[0]
        ref_len = 0
        for k in range(n_cig):
            c = struct.unpack_from("<I", data, p + 36 + l_name + 4 * k)[0]
            if (c & 15) in (0, 2, 3, 7, 8):
                ref_len += c >> 4
        end = start + max(ref_len, 1)
        v0, v1 = voff(p), voff(p + 4 + bs)
        if tid >= 0:
            ch = bins[tid].setdefault(reg2bin_csi(start, end, min_shift, depth), [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            # every bin of every level the record reaches into
            for l in range(depth + 1):
                sh = min_shift + 3 * (depth - l)
                t = ((1 << (3 * l)) - 1) // 7
                for k in range(start >> sh, ((end - 1) >> sh) + 1):
                    loff[tid].setdefault(t + k, v0)
        p += 4 + bs
    out = bytearray(b"CSI\1" + struct.pack("<iii", min_shift, depth, 0) + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            out += struct.pack("<IQi", b, loff[t][b], len(bins[t][b]))
            for v0, v1 in bins[t][b]:
                out += struct.pack("<QQ", v0, v1)
    return bgzf(bytes(out))


def bai(bam_bytes: bytes) -> bytes:
    """A BAI index of a coordinate-sorted BAM as `samtools index` writes it (SAM specification 5.2): bins with their
    chunks (virtual offsets, adjacent chunks of a bin merged), the 16 kb linear index, without the metadata pseudo-bin.
    Virtual offsets of positions at a block boundary are written as "end of the previous block", like htslib's
    bgzf_tell after a record that ends there."""
    import zlib
    # inflate every block, remember where it starts (compressed offset, inflated offset)
    blocks, data, pos = [], bytearray(), 0
    while pos + 18 <= len(bam_bytes):
        bsize = struct.unpack_from("<H", bam_bytes, pos + 16)[0] + 1
        raw = zlib.decompress(bam_bytes[pos + 18:pos + bsize - 8], -15)
        blocks.append((pos, len(data), len(raw)))
        data += raw
        pos += bsize

    import bisect
    nonempty = [b for b in blocks if b[2] > 0]
    starts = [b[1] for b in nonempty]

    def voff(u):   # virtual offset of inflated position u ("end of block" form at boundaries)
        k = max(bisect.bisect_left(starts, u) - 1, 0)      # the last block that starts before u (the first, for u = 0)
        c, s, n = nonempty[k]
        if not s <= u <= s + n:
            raise ValueError(u)
        return (c << 16) | (u - s)

    l_text = struct.unpack_from("<i", data, 4)[0]
    n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]
    p = 12 + l_text
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        p += 8 + l_name
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    while p + 4 <= len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        tid, start = struct.unpack_from("<ii", data, p + 4)
        l_name = data[p + 12]
        n_cig = struct.unpack_from("<H", data, p + 16)[0]
        ref_len = 0
        for k in range(n_cig):
            c = struct.unpack_from("<I", data, p + 36 + l_name + 4 * k)[0]
            if (c & 15) in (0, 2, 3, 7, 8):
                ref_len += c >> 4
        end = start + max(ref_len, 1)
        v0, v1 = voff(p), voff(p + 4 + bs)
        if tid >= 0:
            ch = bins[tid].setdefault(reg2bin(start, end), [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(start >> 14, ((end - 1) >> 14) + 1):
                lin[tid].setdefault(w, v0)
        p += 4 + bs
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            out += struct.pack("<Ii", b, len(bins[t][b]))
            for v0, v1 in bins[t][b]:
                out += struct.pack("<QQ", v0, v1)
        n_intv = max(lin[t]) + 1 if lin[t] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):      # windows no record starts in inherit the next smaller offset (htslib fills them)
            last = lin[t].get(w, last)
            out += struct.pack("<Q", last)
    return bytes(out)
