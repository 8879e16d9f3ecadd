#!/usr/bin/env python3
"""Generates tests/golden/sfs_golden.json.

The reference ships no golden vectors and cannot be built here,
so the vectors are produced by the index-free brute-force model of
ping_pong.cpp:4-49 in oracle/svdss_oracle.c (orc_ping_pong_bruteforce: the
interval-size test replaced by memmem over `contig $ revcomp $ ...`).  Inputs
and expected outputs are data; no reference source is stored.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svdss_amd import synth  # noqa: E402
from tests import oracle_lib as O  # noqa: E402


def case(name, contigs, reads):
    text = O.build_text(contigs)
    out = []
    for r in reads:
        raw, n_ext = O.ping_pong_bruteforce(text, r)
        out.append({"read": synth.to_ascii(r), "sfs": raw, "assembled": O.assemble(raw), "n_ext": n_ext})
    return {"name": name, "contigs": [synth.to_ascii(c) for c in contigs], "reads": out}


def main():
    cases = []
    # 1: SURVEY 8(c)-style: a read with 1 SNP, one 30-bp insertion and one 3-bp deletion,
    #    plus perfect forward / reverse-complement reads (no SFS)
    rng = np.random.default_rng(101)
    ref = synth.make_reference([6000], seed=101, repeat_frac=0.0)
    w = ref[0][1000:1400].copy()
    w[90] = (w[90] % 4) + 1
    ins = rng.integers(1, 5, size=30, dtype=np.uint8)
    r = np.concatenate([w[:200], ins, w[200:330], w[333:]])
    cases.append(case("snp_ins_del", ref, [r, ref[0][2000:2500].copy(), synth.revcomp(ref[0][3000:3300])]))
    # 2: two contigs with diverged repeats and an N run; erroneous ragged reads from a haplotype with SVs
    ref = synth.make_reference([20000, 9000], seed=102, repeat_frac=0.1, n_runs=(40,))
    hap, _ = synth.implant_svs(ref, 4, seed=103, min_len=30, max_len=120)
    flat, offs, _ = synth.simulate_reads(hap, 12, 400, 0.01, seed=104, ragged=True)
    reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(12)]
    cases.append(case("repeats_svs_errors", ref, reads))
    # 3: edge cases: N in read, all-N read vs N-free reference (one SFS per base), 1-base reads,
    #    read reaching position 0 inside an SFS, read equal to a whole contig
    ref = synth.make_reference([3000], seed=105, repeat_frac=0.0)
    a = ref[0][500:700].copy(); a[100] = 5
    b = np.full(25, 5, dtype=np.uint8)
    c = ref[0][10:11].copy()
    d = ref[0][800:1000].copy(); d[3] = (d[3] % 4) + 1
    e = ref[0].copy()
    f = np.array([5], dtype=np.uint8)
    cases.append(case("edges", ref, [a, b, c, d, e, f]))
    # ---- round 4: >= 200 more reads by the brute-force model, by kind
    # 4: both strands, errors, two contigs with repeats
    ref = synth.make_reference([12000, 5000], seed=201, repeat_frac=0.08)
    flat, offs, _ = synth.simulate_reads(ref, 60, 300, 0.01, seed=202, ragged=True)
    cases.append(case("both_strands_errors", ref, [flat[offs[i]:offs[i + 1]].copy() for i in range(60)]))
    # 5: N runs in the reference and in the reads (N is symbol 5, its own complement, and matches N: App. A #6)
    ref = synth.make_reference([8000], seed=203, repeat_frac=0.0, n_runs=(120, 30))
    rng = np.random.default_rng(204)
    npos = np.flatnonzero(ref[0] == 5)
    reads = []
    for k in range(24):
        if k % 3 == 0 and len(npos):          # across a boundary of an N run
            c = int(npos[rng.integers(0, len(npos))])
            a = max(0, c - int(rng.integers(20, 150)))
        else:
            a = int(rng.integers(0, 7600))
        r = ref[0][a:a + int(rng.integers(60, 320))].copy()
        if k % 4 == 1:
            r[rng.integers(0, len(r))] = 5      # an N where the reference has a base
        if k % 4 == 2:
            r = synth.revcomp(r)
        reads.append(r)
    cases.append(case("n_runs", ref, reads))
    # 6: reads shorter than any k-mer table order (1 .. 17 symbols), exact / with one error / absent
    ref = synth.make_reference([5000], seed=205, repeat_frac=0.0)
    reads = []
    for ln in range(1, 18):
        a = int(rng.integers(0, 4900))
        r = ref[0][a:a + ln].copy()
        reads.append(r)
        e = r.copy(); e[ln // 2] = (e[ln // 2] % 4) + 1
        reads.append(e)
    cases.append(case("shorter_than_k", ref, reads))
    # 7: reverse-complement palindromes (W + revcomp(W)) in the reference and in reads: both strands of such a read are
    #    the same string, and the two strand coordinates of the bidirectional index coincide
    parts, pal_reads = [], []
    for k in range(10):
        w = rng.integers(1, 5, size=int(rng.integers(8, 40)), dtype=np.uint8)
        pal = np.concatenate([w, synth.revcomp(w)])
        parts += [rng.integers(1, 5, size=200, dtype=np.uint8), pal]
        left = parts[-2][-30:]
        pal_reads.append(np.concatenate([left, pal]))
        q = np.concatenate([left, pal]).copy(); q[len(left) + len(w)] = (q[len(left) + len(w)] % 4) + 1
        pal_reads.append(q)
    ref = [np.concatenate(parts + [rng.integers(1, 5, size=200, dtype=np.uint8)])]
    cases.append(case("palindromes", ref, pal_reads))
    # 8: the junctions of the text contig $ revcomp $ contig' $ ...: reads that would only occur ACROSS a '$' (the end
    #    of a contig followed by the start of its reverse complement / of the next contig) must not be found; reads whose
    #    suffix (prefix) ends exactly at a contig end (start), both strands, exact and with an error near the end; reads that
    #    run past the end of a contig.  (The forward loop has no test for the read's end: see oracle/svdss_oracle.c on why
    #    it never reaches P[l].)
    ref = synth.make_reference([3000, 1500, 700], seed=206, repeat_frac=0.0)
    reads = []
    for ci, c in enumerate(ref):
        rc = synth.revcomp(c)
        nxt = ref[(ci + 1) % len(ref)]
        for ln in (40, 151):
            reads.append(c[-ln:].copy())                                       # suffix ends at the contig end
            reads.append(c[:ln].copy())                                        # prefix starts at the contig start
            reads.append(synth.revcomp(c[-ln:]))                               # the same on the other strand
            reads.append(synth.revcomp(c[:ln]))
            e = c[-ln:].copy(); e[-3] = (e[-3] % 4) + 1; reads.append(e)       # an error three bases from the end
            e = c[:ln].copy(); e[2] = (e[2] % 4) + 1; reads.append(e)
            reads.append(np.concatenate([c[-ln:], rc[:ln]]))                   # across contig $ revcomp
            reads.append(np.concatenate([rc[-ln:], nxt[:ln]]))                 # across revcomp $ next contig
            reads.append(np.concatenate([c[-ln:], rng.integers(1, 5, size=25, dtype=np.uint8)]))   # past the end
            reads.append(np.concatenate([rng.integers(1, 5, size=25, dtype=np.uint8), c[:ln]]))    # before the start
    cases.append(case("contig_ends_and_junctions", ref, reads))
    path = os.path.join(ROOT, "tests", "golden", "sfs_golden.json")
    with open(path, "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py (orc_ping_pong_bruteforce)", "cases": cases}, fh)
    print("wrote", path, sum(len(c["reads"]) for c in cases), "reads")


if __name__ == "__main__":
    main()
