"""The binary's host-side file readers (BAM / BGZF, BAI, FASTA / FASTQ, rld0 .fmd, the index sidecars) under AddressSanitizer and
UndefinedBehaviorSanitizer on the CPU: tests/host_io_harness.cpp reads a valid file to its end, and hundreds of damaged
ones -- flipped bits and truncations of the compressed file, and records / headers damaged BEFORE compression so that the
damage reaches the parsers behind a valid BGZF layer -- either to a clean end or to an error message.  An out-of-bounds
read, an integer overflow, an exception out of a reader or a hang is a failure."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import bam_writer as W
from tests.common import ROOT

HARNESS = os.path.join(ROOT, "tests", "_host_io_harness")
SRC = [os.path.join(ROOT, "tests", "host_io_harness.cpp"), os.path.join(ROOT, "svdss_amd", "csrc", "rld0.cpp"),
       os.path.join(ROOT, "svdss_amd", "csrc", "index_build.cpp")]
DEPS = SRC + [os.path.join(ROOT, "svdss_amd", "csrc", h) for h in ("bam_reader.h", "bgzf_inflater.h", "bai_index.h", "fastx_reader.h", "rld0.h", "index_host.h", "fmd_layout.h", "sfs_file.h", "bgzf_scanner.h",
                                                                   "bam_device_select.h")]


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(HARNESS) or any(os.path.getmtime(d) > os.path.getmtime(HARNESS) for d in DEPS):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fopenmp", "-pthread", "-o", HARNESS] + SRC + ["-lz", "-ldl"], check=True)
    return HARNESS


def run(harness, mode, path):
    env = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1", OMP_NUM_THREADS="2", SVDSS_GPU_INFLATE="0")
    p = subprocess.run([harness, mode, path], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode in (0, 1), f"{mode} {path}: exit {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    return p.returncode, p.stdout


def _records(rng, n):
    recs, pos = [], 100
    for i in range(n):
        l = int(rng.integers(100, 3000))
        seq = "".join(rng.choice(list("ACGTN"), size=l))
        cigar = [("S", 5), ("M", l - 25), ("I", 10), ("M", 10)] if i % 3 else [("M", l)]
        tags = [("XF", "i", int(i % 4)), ("HP", "C", int(i % 3)), ("RG", "Z", "grp")][: 1 + i % 3]
        recs.append(W.record(f"read{i}", 0 if i % 7 else 256, int(i % 2), pos, 60, cigar, seq, tags))
        pos += int(rng.integers(50, 900))
    return recs


def _plain_bam(refs, recs):
    """the uncompressed BAM stream (header + records) of bam_writer.bam"""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    return hdr, b"".join(recs)


def test_bam_reader_on_damaged_files(harness, tmp_path):
    rng = np.random.default_rng(1)
    refs = [("chr1", 5_000_000), ("chr2", 3_000_000)]
    recs = _records(rng, 300)
    good = W.bam(refs, recs)
    path = str(tmp_path / "x.bam")
    open(path, "wb").write(good)
    rc, out = run(harness, "bam", path)
    assert rc == 0 and "pass 0: 300 records" in out and "pass 1: 300 records" in out
    hdr, body = _plain_bam(refs, recs)
    plain = hdr + body
    n_err = n_ok = 0
    for k in range(160):
        kind = k % 8
        if kind == 0:                                      # bits of the compressed file
            d = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
            data = bytes(d)
        elif kind == 1:                                    # truncated file
            data = good[:int(rng.integers(0, len(good)))]
        elif kind == 2:                                    # a BGZF header field (sizes, magic, subfield)
            d = bytearray(good)
            at = int(rng.integers(0, 18))
            d[at] = int(rng.integers(0, 256))
            data = bytes(d)
        else:                                              # the stream damaged before compression: valid BGZF around it
            d = bytearray(plain)
            if kind == 3:                                  # a record's block_size / core fields
                off = len(hdr)
                for _ in range(int(rng.integers(0, 40))):
                    bs, = struct.unpack_from("<i", d, off)
                    off += 4 + bs
                field = int(rng.choice([0, 4, 12, 16, 20, 24]))      # block_size, refID, l_read_name.., n_cigar.., l_seq
                struct.pack_into("<i", d, off + field, int(rng.choice([-1, -2**31, 2**31 - 1, 0, 1, 31, 2**20, int(rng.integers(-1000, 100000))])))
            elif kind == 4:                                # header lengths
                field = int(rng.choice([4, 8 + struct.unpack_from("<i", d, 4)[0]]))
                struct.pack_into("<i", d, field, int(rng.choice([-1, -2**31, 2**31 - 1, 2**24, 0])))
            elif kind == 5:                                # random bytes anywhere in the records
                for _ in range(int(rng.integers(1, 30))):
                    d[int(rng.integers(len(hdr), len(d)))] = int(rng.integers(0, 256))
            elif kind == 6:                                # aux fields: unknown types, unterminated strings, B arrays
                tail = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
                d = d[:len(d) - 3] + tail
                bs_off = len(hdr)
                last = bs_off
                while bs_off < len(plain):
                    last = bs_off
                    bs_off += 4 + struct.unpack_from("<i", plain, bs_off)[0]
                struct.pack_into("<i", d, last, len(d) - last - 4)
            else:                                          # cut inside the stream
                d = d[:int(rng.integers(0, len(d)))]
            data = W.bgzf(bytes(d), block=int(rng.choice([60000, 700, 65280])))
        open(path, "wb").write(data)
        rc, out = run(harness, "bam", path)
        n_err += rc == 1
        n_ok += rc == 0
    assert n_err > 60 and n_ok > 5                        # both outcomes occur; the sanitizers stayed silent


def test_device_path_host_side_on_damaged_files(harness, tmp_path):
    """Round 4: what the device path keeps on the host -- the BAM header probe, the BGZF member scanner (several loader
    threads, slabs of 64 KB and 1 MB so that members straddle slabs) and the record view over returned bytes -- on the valid
    file and on the same kinds of damage as above."""
    rng = np.random.default_rng(2)
    refs = [("chr1", 5_000_000), ("chr2", 3_000_000)]
    recs = _records(rng, 300)
    good = W.bam(refs, recs)
    path = str(tmp_path / "x.bam")
    open(path, "wb").write(good)
    rc, out = run(harness, "scan", path)
    assert rc == 0 and out.count("300 records") == 2, out
    # round 5: the same file as regions cut at BGZF members (`search --gpus N`): blocks of every size, one that is empty
    rc, out = run(harness, "regions", path)
    assert rc == 0 and out.count("the file's") == 3, out
    odd = str(tmp_path / "odd.bam")
    hdr0, body0 = _plain_bam(refs, recs)
    parts, i, plain0 = [], 0, hdr0 + body0
    while i < len(plain0):
        n = int(rng.integers(1, 9000))
        parts.append(W._bgzf_block(plain0[i:i + n]))
        if rng.random() < 0.05:
            parts.append(W._bgzf_block(b""))
        i += n
    open(odd, "wb").write(b"".join(parts) + W._bgzf_block(b""))
    rc, out = run(harness, "regions", odd)
    assert rc == 0 and out.count("the file's") == 3, out
    hdr, body = _plain_bam(refs, recs)
    plain = hdr + body
    n_err = n_ok = 0
    for k in range(120):
        kind = k % 6
        if kind == 0:
            d = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
            data = bytes(d)
        elif kind == 1:
            data = good[:int(rng.integers(0, len(good)))]
        elif kind == 2:                                    # a BGZF header field of some member (sizes, magic, subfield)
            d = bytearray(good)
            blocks = []
            pos = 0
            while pos + 18 <= len(good):
                blocks.append(pos)
                pos += struct.unpack_from("<H", good, pos + 16)[0] + 1
            at = blocks[int(rng.integers(0, len(blocks)))] + int(rng.integers(0, 18))
            d[at] = int(rng.integers(0, 256))
            data = bytes(d)
        elif kind == 3:                                    # header lengths, behind a valid BGZF layer
            d = bytearray(plain)
            field = int(rng.choice([4, 8 + struct.unpack_from("<i", d, 4)[0], 12 + struct.unpack_from("<i", d, 4)[0]]))
            struct.pack_into("<i", d, field, int(rng.choice([-1, -2**31, 2**31 - 1, 2**24, 0, 7])))
            data = W.bgzf(bytes(d), block=int(rng.choice([60000, 700])))
        elif kind == 4:                                    # record fields
            d = bytearray(plain)
            off = len(hdr)
            for _ in range(int(rng.integers(0, 40))):
                bs, = struct.unpack_from("<i", d, off)
                off += 4 + bs
            struct.pack_into("<i", d, off + int(rng.choice([0, 12, 16, 20])), int(rng.choice([-1, -2**31, 2**31 - 1, 0, 31, 2**20])))
            data = W.bgzf(bytes(d), block=int(rng.choice([60000, 700, 65280])))
        else:                                              # members of odd sizes, empty members in between
            parts = []
            i = 0
            while i < len(plain):
                n = int(rng.integers(1, 4000))
                parts.append(W._bgzf_block(plain[i:i + n]))
                if rng.random() < 0.1:
                    parts.append(W._bgzf_block(b""))
                i += n
            parts.append(W._bgzf_block(b""))
            data = b"".join(parts)
        open(path, "wb").write(data)
        rc, out = run(harness, "scan", path)
        n_err += rc == 1
        n_ok += rc == 0
        run(harness, "regions", path)                      # (an error or the file's members; never something else)
    assert n_err > 40 and n_ok > 15


def test_bai_and_region_scan_on_damaged_files(harness, tmp_path):
    rng = np.random.default_rng(2)
    refs = [("chr1", 5_000_000), ("chr2", 3_000_000)]
    recs = [W.record(f"r{i}", 0, i % 2, 1000 * i, 60, [("M", 500)], "ACGT" * 125) for i in range(400)]
    recs.sort(key=lambda r: (struct.unpack_from("<i", r, 4)[0], struct.unpack_from("<i", r, 8)[0]))
    good = W.bam(refs, recs)
    bai = W.bai(good)
    path = str(tmp_path / "y.bam")
    open(path, "wb").write(good)
    open(path + ".bai", "wb").write(bai)
    rc, out = run(harness, "bai", path)
    assert rc == 0 and "records" in out
    for k in range(80):
        b = bytearray(bai)
        x = k % 4
        if x == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif x == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif x == 2:                                       # counts: n_ref, n_bin, n_chunk, n_intv
            struct.pack_into("<i", b, int(rng.choice([4, 8, 16])), int(rng.choice([-1, 2**31 - 1, 2**20, 0])))
        else:                                              # chunk offsets that point anywhere
            at = int(rng.integers(8, len(b) - 8))
            struct.pack_into("<Q", b, at, int(rng.integers(0, 2**48)))
        open(path + ".bai", "wb").write(bytes(b))
        run(harness, "bai", path)
    # a good index over a damaged file
    open(path + ".bai", "wb").write(bai)
    for k in range(30):
        d = bytearray(good)
        d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
        open(path, "wb").write(bytes(d[:int(rng.integers(len(d) // 2, len(d) + 1))]))
        run(harness, "bai", path)


def test_fastx_reader_on_odd_files(harness, tmp_path):
    rng = np.random.default_rng(3)
    path = str(tmp_path / "z.fa")
    cases = [b"", b">", b">a", b">a\n", b"@q\nACGT\n+\n", b"@q\nACGT\n+\nII", b"ACGT\n>x\nAC\r\nGT\r\n", b">x\n" + b"A" * 5_000_000,
             b"\n\n\n>e\n\n\n>f\nAC\n", b"@a\nAC\nGT\n+\nII\nII\n@b\nA\n+\n@\n", bytes(rng.integers(0, 256, size=100000, dtype=np.uint8))]
    for c in cases:
        for gz in (False, True):
            open(path, "wb").write(gzip.compress(c) if gz else c)
            rc, out = run(harness, "fastx", path)
            assert rc == 0
    # the mapped loader of `SVDSS call` takes plain FASTA with '\n' line ends -- and then gives what the line reader gives
    big = b"stray line\n>chr1 some description\n" + b"\n".join(bytes(rng.choice(np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), size=int(w))) for w in rng.integers(0, 90, size=60000))
    big += b"\n>chr2\tx\n\n\nACGT\n>empty\n>chr1\nTTTT" + b"\n>long\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=20_000_000))
    open(path, "wb").write(big)
    rc, out = run(harness, "fastx", path)
    assert rc == 0 and "mapped loader: the same 5 records" in out, out
    for odd in (b">a\nAC\r\nGT\n", b"@q\nACGT\n+\nIIII\n", b">a\nAC\n@b\nGT\n", gzip.compress(b">a\nACGT\n"), b"no header at all\n"):
        open(path, "wb").write(odd)
        rc, out = run(harness, "fastx", path)
        assert rc == 0 and "mapped loader: declined" in out, out
    # a gzip stream cut short / damaged
    blob = gzip.compress(b">x\n" + b"ACGT" * 100000 + b"\n")
    for cut in (10, len(blob) // 2, len(blob) - 3):
        open(path, "wb").write(blob[:cut])
        run(harness, "fastx", path)


def test_rld0_reader_on_damaged_files(harness, tmp_path, monkeypatch):
    import svdss_amd
    from svdss_amd import synth
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    ref = synth.make_reference([20000, 3000], seed=4, n_runs=(50,))
    path = str(tmp_path / "i.fmd")
    svdss_amd.FMDIndex.build(ref).save_fmd(path)
    good = open(path, "rb").read()
    rc, out = run(harness, "fmd", path)
    assert rc == 0 and "4 strings" in out
    rng = np.random.default_rng(5)
    n_err = 0
    for k in range(120):
        d = bytearray(good)
        x = k % 5
        if x == 0:
            for _ in range(int(rng.integers(1, 5))):
                d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
        elif x == 1:
            d = d[:int(rng.integers(0, len(d)))]
        elif x == 2:                                       # header: asize / sbits, data length, frames, symbol counts
            at = int(rng.choice([4, 8, 16, 24, 32, 40, 64]))
            struct.pack_into("<Q", d, at, int(rng.choice([0, 1, 2**63, 2**40, 2**32, int(rng.integers(0, 2**20))])))
        elif x == 3:                                       # block headers: type bits and counts
            at = 72 + 64 * int(rng.integers(0, (len(d) - 200) // 64))
            struct.pack_into("<Q", d, at, int(rng.integers(0, 2**63)) | (int(rng.integers(0, 4)) << 62))
        else:
            at = int(rng.integers(72, len(d) - 8))
            d[at:at + 8] = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))
        open(path, "wb").write(bytes(d))
        rc, out = run(harness, "fmd", path)
        n_err += rc == 1
    assert n_err > 30


def test_index_sidecars_on_damaged_files(harness, tmp_path, monkeypatch):
    import svdss_amd
    from svdss_amd import synth
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    ref = synth.make_reference([9000, 2000], seed=6, n_runs=(30,))
    ix = svdss_amd.FMDIndex.build(ref)
    rec, full = str(tmp_path / "r.svdss"), str(tmp_path / "f.svdss")
    ix.save_records(rec)
    ix.save(full)
    rng = np.random.default_rng(7)
    for path in (rec, full):
        good = open(path, "rb").read()
        rc, out = run(harness, "sidecar", path)
        assert rc == 0 and ("records: 11000 bases" in out or "full layout" in out)
        n_err = 0
        for k in range(60):
            d = bytearray(good)
            x = k % 4
            if x == 0:                                     # header fields: lengths, counts, totals
                at = 8 * int(rng.integers(1, 12))
                struct.pack_into("<q", d, at, int(rng.choice([-1, 0, 1, 2**62, 2**40, 2**31, int(rng.integers(0, 2**20))])))
            elif x == 1:
                d = d[:int(rng.integers(0, len(d)))]
            elif x == 2:                                   # payload bytes: symbols out of range, suffix-array entries anywhere
                for _ in range(int(rng.integers(1, 20))):
                    d[int(rng.integers(80, len(d)))] = int(rng.integers(0, 256))
            else:
                at = int(rng.integers(8, len(d) - 8))
                d[at:at + 8] = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))
            open(path, "wb").write(bytes(d))
            rc, out = run(harness, "sidecar", path)
            n_err += rc == 1
        assert n_err > 20


def test_sfs_file_readers_agree(harness, tmp_path):
    """csrc/sfs_file.h: the piecewise .sfs reader of `SVDSS call` against the line-by-line one it replaces, on the text
    `SVDSS search` writes and on every kind of line the format does not have."""
    rng = np.random.default_rng(8)
    path = str(tmp_path / "x.sfs")

    def good(n_reads):
        out = []
        for i in range(n_reads):
            name = f"m64011_190830_220126/{int(rng.integers(0, n_reads // 2 + 2))}/ccs"     # (names repeat: the later list wins)
            for k in range(int(rng.integers(1, 90))):
                out.append(f"{name if k == 0 else '*'}\t{int(rng.integers(0, 20000))}\t{int(rng.integers(1, 300))}\t{int(rng.integers(0, 3))}\t\n")
        return out
    lines = good(30000)
    open(path, "w").write("".join(lines))                      # 1.3 M lines, ~18 MB: several pieces per thread count
    rc, out = run(harness, "sfs", path)
    assert rc == 0 and "reads" in out
    odd = ["*\t1\t2\t0\t\n", "\n", "   \n", "name only\n", "r1\t5\n", "r2\t5\t6\n", "r3\t5\t6\t7\textra\tfields\n", "r4 5 6 7\n",
           "r5\t-5\t+6\t7\n", "r6\t5x\t6\t7\n", "r7\t5\t6\t7x\n", "*\t9\t9\t9\r\n", "r8\t\t5\t\t6\t\t7\n", "*\n", "* * * *\n",
           "r9\t99999999999\t1\t1\n", "r9b\t1\t-99999999999\t1\n", "r9c\t2147483647\t-2147483648\t2147483648\n", "r" * 4095 + "\t1\t2\t3\n", "q" * 4096 + "\t1\t2\t3\n", "\tr10\t1\t2\t3\n", "r11\t1\t2\t3"]
    for trial in range(12):
        body = lines[:int(rng.integers(0, 3000))]
        for _ in range(int(rng.integers(1, 60))):
            body.insert(int(rng.integers(0, len(body) + 1)), odd[int(rng.integers(0, len(odd)))])
        if trial % 4 == 1:
            body.insert(len(body) // 2, "L" * 9000 + "\t1\t2\t3\n")              # a line fgets would split
        if trial % 4 == 2:
            body = ["*\t4\t4\t4\t\n"] * 3 + body                                  # '*' before any name
        data = "".join(body)
        if trial % 4 == 3:
            data = data[:int(rng.integers(0, len(data) + 1))]                     # cut anywhere
        open(path, "w").write(data)
        rc, out = run(harness, "sfs", path)
        assert rc == 0, out
    open(path, "wb").write(bytes(rng.integers(0, 256, size=300000, dtype=np.uint8)))   # not a text file at all
    rc, out = run(harness, "sfs", path)
    assert rc == 0, out
    open(path, "w").write("")
    assert run(harness, "sfs", path)[0] == 0
