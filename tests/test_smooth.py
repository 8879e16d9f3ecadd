"""`SVDSS smooth` (C++ CLI, csrc/smooth_host.cpp) against the Python mirror of smoother.cpp: same
accuracy threshold, same XF tags, same rewritten SEQ/QUAL/CIGAR, same record order, dropped records."""
import os
import subprocess

import numpy as np

from svdss_amd import bamio, smoother, synth
from svdss_amd.clusterer import Alignment
from tests import bam_writer
from tests.common import ROOT
from tests.pipeline_sim import add_errors, simulate

BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")
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
