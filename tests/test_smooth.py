"""`SVDSS smooth` (C++ CLI, csrc/smooth_host.cpp) against the Python mirror of smoother.cpp: same
accuracy threshold, same XF tags, same rewritten SEQ/QUAL/CIGAR, same record order, dropped records."""
import os
import subprocess

import numpy as np

from svdss_amd import synth
from tests.mirror import bamio, smoother
from tests.mirror.clusterer import Alignment
from tests import bam_writer
from tests.common import ROOT
from tests.pipeline_sim import add_errors, simulate

from tests.common import BIN  # noqa: E402
OPS = {"M": 0, "I": 1, "D": 2, "S": 4}


def test_smooth_cli_matches_mirror(tmp_path):
    ref, svs, reads = simulate(ref_lens=(80000, 30000), n_svs=5, coverage=8, read_len=3000, seed=8)
    rng = np.random.default_rng(2)
    names = ["c0", "c1"]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for n, c in zip(names, ref):
            fh.write(f">{n}\n{synth.to_ascii(c).lower()}\n")       # lower case: load_chromosomes upper-cases
    recs, alns = [], []
    for k, (n, tid, pos, cig, seq, hp) in enumerate(reads):
        err = 0.03 if k % 17 == 0 else 0.008                          # a few dirty reads -> XF=1
        s2, c2 = add_errors(seq, cig, rng, err)
        qual = bytes(rng.integers(1, 60, size=len(s2)).astype(np.uint8))
        flag = 256 if k % 23 == 5 else (16 if k % 2 else 0)
        mapq = 5 if k % 29 == 7 else 60
        tags = [("NM", "i", 7)] + ([("XF", "i", 3)] if k % 5 == 0 else []) + [("RG", "Z", "x")]
        recs.append(bam_writer.record(n, flag, tid, pos, mapq, c2, s2, tags, qual))
        alns.append(Alignment(n, flag, tid, pos, mapq, [(l, OPS[o]) for o, l in c2], s2, {}, qual))
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    out = tmp_path / "smoothed.bam"
    with open(out, "wb") as fh:
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "3"], stdout=fh,
                           stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    ref_names, ref_lens, got = bamio.read_bam(str(out))
    assert ref_names == names and ref_lens == [len(c) for c in ref]
    chromosomes = {n: synth.to_ascii(c) for n, c in zip(names, ref)}
    acc = smoother.compute_maxaccuracy(alns, names, chromosomes)
    expect = []
    for a in alns:
        if not smoother.eligible(a, names, chromosomes, 20):
            continue
        xf, ns, nq, nc = smoother.smooth_read(a, a.qual, chromosomes[names[a.tid]], acc)
        expect.append((a, xf, ns, nq, nc))
    assert len(got) == len(expect) < len(alns)                       # secondary / low-mapq records were dropped
    xfs = []
    for g, (a, xf, ns, nq, nc) in zip(got, expect):
        assert g.qname == a.qname and g.pos == a.pos and g.flag == a.flag and g.tags["XF"] == xf and g.tags["NM"] == 7
        if xf == 0:
            assert g.seq == ns and g.qual == nq and g.cigar == nc
            # smoothed read == reference on every aligned stretch
            rp, qp = g.pos, 0
            for l, op in g.cigar:
                if op == 0:
                    assert g.seq[qp:qp + l] == chromosomes[names[g.tid]][rp:rp + l]
                    rp += l; qp += l
                elif op == 1 or op == 4:
                    assert l > 20 or op == 4
                    qp += l
                elif op == 2:
                    assert l > 20
                    rp += l
        else:
            assert g.seq == a.seq and g.qual == a.qual and g.cigar == a.cigar
        xfs.append(xf)
    assert xfs.count(0) > 5 and xfs.count(1) >= 1 and xfs.count(2) > 5


def test_smooth_output_does_not_depend_on_threads_and_spans_many_bgzf_chunks(tmp_path):
    """The parallel BGZF reader (512-block chunks, read ahead) and writer (256-block batches) must give the same
    bytes whatever the thread count, on a BAM large enough to span several chunks."""
    import struct
    import zlib

    rng = np.random.default_rng(7)
    ref = rng.integers(1, 5, size=300000).astype(np.uint8)
    fa = tmp_path / "ref.fa"
    fa.write_text(">c1\n" + synth.to_ascii(ref) + "\n")
    code = np.array([0, 1, 2, 4, 8, 15], dtype=np.uint8)            # nt6 -> BAM 4-bit
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:%d\n" % len(ref)
    data = [b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1)
            + struct.pack("<i", 3) + b"c1\0" + struct.pack("<i", len(ref))]
    n_reads, ln = 9000, 4000
    starts = np.sort(rng.integers(0, len(ref) - ln, size=n_reads))
    for i, st in enumerate(starts):
        seq = ref[st:st + ln].copy()
        e = rng.random(ln) < 0.004
        seq[e] = (seq[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        c = code[seq]
        packed = ((c[0::2] << 4) | c[1::2]).tobytes()
        name = ("r%06d" % i).encode() + b"\0"
        core = struct.pack("<iiBBHHHiiii", 0, int(st), len(name), 60, 4680, 1, 0, ln, -1, -1, 0)
        body = core + name + struct.pack("<I", ln << 4) + packed + rng.integers(20, 60, size=ln, dtype=np.uint8).tobytes()
        data.append(struct.pack("<i", len(body)) + body)
    raw = b"".join(data)
    assert len(raw) > 600 * 65280                                   # more than one 512-block chunk

    def block(d):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        cd = co.compress(d) + co.flush()
        return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cd) + 25) + cd
                + struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d)))

    bam = tmp_path / "in.bam"
    with open(bam, "wb") as fh:
        for k in range(0, len(raw), 65280):
            fh.write(block(raw[k:k + 65280]))
        fh.write(block(b""))
    outs = []
    for threads in ("1", "6"):
        out = tmp_path / f"out{threads}.bam"
        with open(out, "wb") as fh:
            r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", threads], stdout=fh,
                               stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] and len(outs[0]) > 0
    # every way the BGZF loader gets at the file gives the same stream: per-loader pread of file ranges (default; with
    # ranges as small as one block, so that block chains cross many range boundaries and some ranges hold no block start),
    # sequential reads (pipes), a mapping of the file; few or many chunks in flight
    for k, env in enumerate(({"SVDSS_BAM_STREAM": "1"}, {"SVDSS_BAM_MMAP": "1"}, {"SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_AHEAD": "3"},
                             {"SVDSS_BAM_SLAB_KB": "100"}, {"SVDSS_BAM_SLAB_KB": "1000", "SVDSS_BAM_AHEAD": "40"})):
        out = tmp_path / f"out_v{k}.bam"
        with open(out, "wb") as fh:
            r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4"], stdout=fh,
                               stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, (env, r.stderr)
        assert out.read_bytes() == outs[0], env
    # a file cut in the middle of a block is an error, not a shorter stream
    cut = tmp_path / "cut.bam"
    cut.write_bytes(bam.read_bytes()[:os.path.getsize(bam) // 2])
    for env in ({}, {"SVDSS_BAM_SLAB_KB": "100"}, {"SVDSS_BAM_STREAM": "1"}):
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(cut), "--threads", "4"], stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 1 and b"truncated" in r.stderr, (env, r.stderr[-300:])
    names, lens, alns = bamio.read_bam(str(tmp_path / "out6.bam"))
    assert len(alns) == n_reads and names == ["c1"]


def test_inconsistent_records_pass_through_with_xf3(tmp_path):
    """an alignment that overhangs the contig end, or whose CIGAR does not add up to l_seq, used to be walked
    past the end of the reference / read (invalid BAM written silently).  Such a record now leaves `SVDSS smooth` as it
    came in, tagged XF = 3 (the reference's tag for a record it could not rebuild, smoother.cpp:219-228); the valid
    records around it are smoothed as usual."""
    rng = np.random.default_rng(4)
    ref = synth.to_ascii(rng.integers(1, 5, size=3000).astype(np.uint8))
    fa = tmp_path / "ref.fa"
    fa.write_text(f">c0\n{ref}\n")
    good = ref[1000:1070] + ref[1100:1180]                   # 30 bases deleted after the 70th
    over = ref[2900:3000] + "ACGT" * 13                      # 152 bases "aligned" at 2900 with 152M: 52 past the contig end
    short = ref[500:600]                                     # 100 bases with a 120M CIGAR
    recs = [bam_writer.record("short", 0, 0, 500, 60, [("M", 120)], short, [], bytes([30] * 100)),
            bam_writer.record("good", 0, 0, 1000, 60, [("M", 70), ("D", 30), ("M", 80)], good, [], bytes([30] * 150)),
            bam_writer.record("over", 0, 0, 2900, 60, [("M", 152)], over, [], bytes([30] * 152))]
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([("c0", 3000)], recs))
    out = tmp_path / "out.bam"
    with open(out, "wb") as fh:
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "2"], stdout=fh,
                           stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    _, _, got = bamio.read_bam(str(out))
    by = {a.qname: a for a in got}
    assert by["over"].tags["XF"] == 3 and by["over"].seq == over and by["over"].cigar == [(152, 0)]
    assert by["short"].tags["XF"] == 3 and by["short"].seq == short and by["short"].cigar == [(120, 0)]
    assert by["good"].tags["XF"] == 0 and by["good"].cigar == [(70, 0), (30, 2), (80, 0)]


def test_smooth_output_does_not_depend_on_how_the_fasta_is_written(tmp_path):
    """load_chromosomes (chromosomes.cpp:9-27) upper-cases and joins lines whatever the file looks like.  `smooth` reads a
    plain FASTA through a mapping with several threads (fastx_reader.h load_fasta_mapped) and everything else -- gzip,
    CRLF line ends -- line by line: one line per chromosome, 60-column lines in mixed case, the same with CRLF, gzipped,
    and the line reader forced (SVDSS_FASTA_SERIAL) must all give the same bytes."""
    import gzip
    ref, svs, reads = simulate(ref_lens=(50000, 20011), n_svs=3, coverage=4, read_len=2500, seed=21)
    rng = np.random.default_rng(5)
    names = ["chrA", "chrB"]
    recs = []
    for k, (n, tid, pos, cig, seq, hp) in enumerate(reads):
        s2, c2 = add_errors(seq, cig, rng, 0.01)
        recs.append(bam_writer.record(n, 16 if k % 2 else 0, tid, pos, 60, c2, s2, [], bytes(rng.integers(1, 60, size=len(s2)).astype(np.uint8))))
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    texts = [synth.to_ascii(c) for c in ref]

    def mixed(s, k):
        return "".join(ch.lower() if (i * 7 + k) % 3 == 0 else ch for i, ch in enumerate(s))

    def body(width, eol, case):
        out = []
        for k, (n, t) in enumerate(zip(names, texts)):
            out.append(f">{n} some description{eol}")
            t = mixed(t, k) if case else t
            if width:
                out.extend(t[i:i + width] + eol for i in range(0, len(t), width))
            else:
                out.append(t + eol)
        return "".join(out).encode()

    forms = {"one_line.fa": body(0, "\n", False), "wrapped_mixed.fa": body(60, "\n", True), "crlf.fa": body(60, "\r\n", True),
             "odd_width.fa": body(61, "\n", True)}
    outs = {}
    for name, data in forms.items():
        p = tmp_path / name
        p.write_bytes(data)
        for env in ({}, {"SVDSS_FASTA_SERIAL": "1"}):
            r = subprocess.run([BIN, "smooth", "--reference", str(p), "--bam", str(bam), "--threads", "5"], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()
            outs[(name, bool(env))] = r.stdout
    gz = tmp_path / "wrapped.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(forms["wrapped_mixed.fa"])
    r = subprocess.run([BIN, "smooth", "--reference", str(gz), "--bam", str(bam), "--threads", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    outs[("gz", False)] = r.stdout
    first = outs[("one_line.fa", False)]
    assert len(first) > 10000 and all(v == first for v in outs.values())
