"""Developer benchmark: the `SVDSS index` / `SVDSS search` binaries end to end on a synthetic FASTA + BAM
(whole process wall time: BGZF inflate, record parsing, GPU search, text output).  Runs on the GPU box.

  python tools/e2e_search.py [ref_bp] [n_reads] [read_len] [workdir]
E2E_REPEAT=k writes the record blocks k times; E2E_QUAL=binned draws the qualities from seven values (default: 40).
"""
import json
import multiprocessing as mp
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE16 = np.array([1, 2, 4, 8], dtype=np.uint8)       # A C G T in BAM's 4-bit alphabet


def _bgzf_block(data):
    comp = zlib.compressobj(1, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25)
    return hdr + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def write_bam(path, ref_name, ref, n_reads, read_len, seed=5, err=0.005, repeat=1):
    rng = np.random.default_rng(seed)
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (ref_name, len(ref))
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1)
    hdr += struct.pack("<i", len(ref_name) + 1) + ref_name.encode() + b"\0" + struct.pack("<i", len(ref))
    starts = np.sort(rng.integers(0, len(ref) - read_len, size=n_reads))
    chunks, cur, cur_len = [], [], 0
    for i, st in enumerate(starts):
        seq = ref[st:st + read_len].copy()
        e = rng.random(read_len) < err
        seq[e] = (seq[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
        c = CODE16[seq]
        if read_len & 1:
            c = np.append(c, np.uint8(0))
        packed = ((c[0::2] << 4) | c[1::2]).tobytes()
        if os.environ.get("E2E_QUAL", "random") == "binned":   # seven quality values, as the sequencers of HiFi reads bin them
            qual = np.array([3, 10, 17, 22, 27, 33, 40], dtype=np.uint8)[rng.integers(0, 7, size=read_len)].tobytes()
        else:
            qual = rng.integers(20, 60, size=read_len, dtype=np.uint8).tobytes()
        name = ("read%07d" % i).encode() + b"\0"
        core = struct.pack("<iiBBHHHiiii", 0, int(st), len(name), 60, 4680, 1, 0, read_len, -1, -1, 0)
        body = core + name + struct.pack("<I", read_len << 4) + packed + qual
        rec = struct.pack("<i", len(body)) + body
        cur.append(rec)
        cur_len += len(rec)
        if cur_len > (8 << 20):
            chunks.append(b"".join(cur)); cur, cur_len = [], 0
    chunks.append(b"".join(cur))
    data = b"".join(chunks)
    blocks = [data[i:i + 65280] for i in range(0, len(data), 65280)]
    with mp.Pool(min(64, os.cpu_count() or 1)) as pool:
        comp = pool.map(_bgzf_block, blocks, chunksize=64)
    # the header in BGZF blocks of its own, then the record blocks `repeat` times (blocks are independent deflate
    # streams: a long input for the price of a short one; read names repeat at a distance of n_reads)
    with open(path, "wb") as f:
        for i in range(0, len(hdr), 65280):
            f.write(_bgzf_block(hdr[i:i + 65280]))
        for _ in range(repeat):
            for b in comp:
                f.write(b)
        f.write(_bgzf_block(b""))
    return len(hdr) + len(data) * repeat


def main():
    ref_bp = int(sys.argv[1]) if len(sys.argv) > 1 else 64444167
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 43000
    read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
    work = sys.argv[4] if len(sys.argv) > 4 else "/tmp/svdss_e2e"
    repeat = int(os.environ.get("E2E_REPEAT", "1"))     # the record blocks written this many times
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(1)
    ref = rng.integers(0, 4, size=ref_bp, dtype=np.uint8)
    fa = os.path.join(work, "ref.fa")
    with open(fa, "wb") as f:
        f.write(b">chrS\n")
        f.write(np.frombuffer(b"ACGT", dtype=np.uint8)[ref].tobytes())
        f.write(b"\n")
    t0 = time.time()
    raw = write_bam(os.path.join(work, "reads.bam"), "chrS", ref, n_reads, read_len, repeat=repeat)
    n_reads *= repeat
    t_gen = time.time() - t0
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    out = {"ref_bp": ref_bp, "n_reads": n_reads, "read_len": read_len, "bam_bytes": os.path.getsize(os.path.join(work, "reads.bam")),
           "bam_uncompressed_bytes": raw, "generate_s": round(t_gen, 2)}
    t0 = time.time()
    subprocess.run([exe, "index", "-t", str(os.cpu_count()), "-d", fa, "-o", os.path.join(work, "ref.fmd")], check=True)
    out["index_s"] = round(time.time() - t0, 2)
    for rep in range(2):
        t0 = time.time()
        with open(os.path.join(work, "sfs.txt"), "wb") as f:
            subprocess.run([exe, "search", "--index", os.path.join(work, "ref.fmd"), "--bam", os.path.join(work, "reads.bam"),
                            "--noputative"] + sys.argv[5:], check=True, stdout=f)
        dt = time.time() - t0
        out["search_s_run%d" % rep] = round(dt, 2)
        out["search_reads_per_s_run%d" % rep] = round(n_reads / dt, 1)
    out["sfs_bytes"] = os.path.getsize(os.path.join(work, "sfs.txt"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
