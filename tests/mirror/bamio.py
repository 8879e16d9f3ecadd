"""Minimal BAM reader for the Python mirrors of the host logic (test infrastructure) (BGZF = concatenated gzip members, so the
standard gzip module inflates it).  The C++ CLI uses csrc/bam_reader.h; this one also decodes
CIGAR and SEQ, which `call` needs."""
import gzip
import struct
from typing import List, Tuple

from .clusterer import Alignment

SEQ16 = "=ACMGRSVTWYHKDBN"


def read_bam(path: str) -> Tuple[List[str], List[int], List[Alignment]]:
    with gzip.open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"BAM\1":
        raise ValueError(f"{path}: not a BAM file")
    o = 4
    (l_text,) = struct.unpack_from("<i", data, o)
    o += 4 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    names, lens = [], []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        o += 4
        names.append(data[o:o + l_name - 1].decode())
        o += l_name
        (l_ref,) = struct.unpack_from("<i", data, o)
        o += 4
        lens.append(l_ref)
    alns = []
    while o < len(data):
        (bs,) = struct.unpack_from("<i", data, o)
        o += 4
        rec = data[o:o + bs]
        o += bs
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        p = 32
        qname = rec[p:p + l_name - 1].decode()
        p += l_name
        cigar = []
        for k in range(n_cig):
            (v,) = struct.unpack_from("<I", rec, p + 4 * k)
            cigar.append((v >> 4, v & 0xf))
        p += 4 * n_cig
        packed = rec[p:p + (l_seq + 1) // 2]
        seq = "".join(SEQ16[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xf] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = bytes(rec[p:p + l_seq])
        p += l_seq
        tags = {}
        while p + 3 <= len(rec):
            tag, ty = rec[p:p + 2].decode(), chr(rec[p + 2])
            p += 3
            if ty in "cCsSiI":
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[ty]
                (val,) = struct.unpack_from("<" + fmt, rec, p)
                p += struct.calcsize(fmt)
                tags[tag] = val
            elif ty == "A":
                p += 1
            elif ty == "f":
                p += 4
            elif ty in "ZH":
                e = rec.index(b"\0", p)
                p = e + 1
            elif ty == "B":
                st = chr(rec[p])
                (cnt,) = struct.unpack_from("<i", rec, p + 1)
                p += 5 + cnt * (1 if st in "cC" else 2 if st in "sS" else 4)
            else:
                break
        alns.append(Alignment(qname, flag, tid, pos, mapq, cigar, seq, tags, qual))
    return names, lens, alns


from svdss_amd.bgzf import bgzf_blocks, gpu_inflate  # noqa: E402,F401  (kept importable from here)
