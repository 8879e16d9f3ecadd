"""tools/chain_dataset.cpp -- the generator of the end-to-end chain's data set (bench.py `e2e_chain_30x`; BENCH TOOLING, not
product): its BAM is what it says it is.  Every record's CIGAR reproduces the read from the reference with errors at the
stated rates (sub : ins : del = 2 : 1.5 : 1.5), the file is sorted, the BAI names every record that overlaps a region
(through the product's own index reader, csrc/bai_index.h), the output does not depend on the number of threads, and a
FASTA that somebody else wrote gives the same reads as the generator's own."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.common import ROOT
from tools import e2e_call_wg as W

SEQ = "=ACMGRSVTWYHKDBN"


def _records(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    lt, = struct.unpack_from("<i", raw, 4)
    o = 8 + lt
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    lens = []
    for _ in range(n_ref):
        ln, = struct.unpack_from("<i", raw, o)
        o += 4 + ln
        lens.append(struct.unpack_from("<i", raw, o)[0])
        o += 4
    recs = []
    while o < len(raw):
        bs, = struct.unpack_from("<i", raw, o)
        tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, _, _, _ = struct.unpack_from("<iiBBHHHiiii", raw, o + 4)
        p = o + 36
        name = raw[p:p + l_name - 1].decode()
        p += l_name
        cig = struct.unpack_from("<%dI" % n_cig, raw, p)
        p += 4 * n_cig
        seq = np.frombuffer(raw, dtype=np.uint8, count=(l_seq + 1) // 2, offset=p)
        p += (l_seq + 1) // 2
        assert raw[p:p + l_seq] == b"\xff" * l_seq
        p += l_seq
        recs.append((name, tid, pos, cig, seq, raw[p:o + 4 + bs], mapq, flag, bin_))
        o += 4 + bs
    return lens, recs


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    exe = W.build_generator()
    d = tmp_path_factory.mktemp("chain")
    subprocess.run([exe, str(d), "3000", "40", "0.002", "0.005", "4", "zlib1"], check=True, capture_output=True)
    return exe, d


def test_cigars_reproduce_the_reads_at_the_stated_error_rates(data):
    exe, d = data
    fa = open(d / "ref.fa").read().split(">")[1:]
    ref = [c.split("\n", 1)[1].replace("\n", "") for c in fa]
    lens, recs = _records(d / "reads.bam")
    assert lens == [len(r) for r in ref] and len(recs) == 3000
    assert [(r[1], r[2]) for r in recs] == sorted((r[1], r[2]) for r in recs)            # coordinate-sorted
    sub = ins = dele = tot = 0
    for name, tid, pos, cig, seq, aux, mapq, flag, bin_ in recs[::7]:
        s = np.empty(2 * len(seq), dtype=np.uint8)
        s[0::2] = seq >> 4
        s[1::2] = seq & 15
        rd = "".join(SEQ[x] for x in s[:15000])
        r, rp, qp = ref[tid], pos, 0
        assert mapq == 60 and flag == 0 and aux == b"" and len(cig) < 65536
        assert (cig[0] & 15) in (0, 4) and (cig[-1] & 15) in (0, 4)
        for c in cig:
            op, l = c & 15, c >> 4
            assert l > 0
            if op == 0:
                a = np.frombuffer(rd[qp:qp + l].encode(), np.uint8)
                b = np.frombuffer(r[rp:rp + l].encode(), np.uint8)
                sub += int((a != b).sum())
                qp += l
                rp += l
            elif op == 1:
                ins += l < 50
                qp += l
            elif op == 4:
                qp += l
            elif op == 2:
                dele += l < 50
                rp += l
            else:
                raise AssertionError(op)
        assert qp == 15000
        end = rp
        # reg2bin of [pos, end)
        b, e = pos, end - 1
        want = next((off + (b >> sh) for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)) if b >> sh == e >> sh), 0)
        assert bin_ == want
        tot += 15000
    assert abs(sub / tot - 0.002) < 0.0004 and abs(ins / tot - 0.0015) < 0.0004 and abs(dele / tot - 0.0015) < 0.0004
    truth = [l.split() for l in open(d / "truth.tsv")]
    assert len(truth) >= 38 and {t[2] for t in truth} == {"INS", "DEL"} and all(50 <= int(t[3]) <= 2000 for t in truth)


def test_bai_names_every_overlapping_record_and_threads_do_not_matter(data, tmp_path):
    exe, d = data
    scan = os.path.join(ROOT, "tests", "native", "_bai_scan")
    src = os.path.join(ROOT, "tests", "native", "bai_scan.cpp")
    if not os.path.exists(scan) or os.path.getmtime(scan) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", scan, src, "-lz", "-ldl"], check=True)
    lens, recs = _records(d / "reads.bam")
    ends = [pos + sum(c >> 4 for c in cig if (c & 15) in (0, 2)) for _, _, pos, cig, *_ in recs]
    rng = np.random.default_rng(3)
    for _ in range(12):
        regs = []
        for _ in range(int(rng.integers(1, 5))):
            t = int(rng.integers(0, len(lens)))
            b = int(rng.integers(0, lens[t]))
            regs.append((t, b, min(lens[t], b + int(rng.choice([1, 300, 20000, 400000])))))
        out = subprocess.run([scan, str(d / "reads.bam"), str(d / "reads.bam.bai")] + ["%d:%d-%d" % r for r in regs], capture_output=True, text=True, check=True).stdout
        got = [l.split("\t")[0] for l in out.splitlines()]
        want = [r[0] for r, e in zip(recs, ends) if any(r[1] == t and r[2] < re_ and e > rb for t, rb, re_ in regs)]
        assert [g for g in got if g in set(want)] == want
    # one thread, and the reads drawn from the FASTA the first run wrote: the same bytes
    for extra in (["1", "zlib1", "0"], ["3", "zlib1", "2"]):
        d2 = tmp_path / ("t" + extra[0])
        d2.mkdir()
        if extra[2] == "2":
            os.symlink(d / "ref.fa", d2 / "ref.fa")
        subprocess.run([exe, str(d2), "3000", "40", "0.002", "0.005"] + extra, check=True, capture_output=True)
        assert (d2 / "reads.bam").read_bytes() == (d / "reads.bam").read_bytes()
        assert (d2 / "reads.bam.bai").read_bytes() == (d / "reads.bam.bai").read_bytes()
