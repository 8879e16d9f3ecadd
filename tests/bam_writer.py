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
