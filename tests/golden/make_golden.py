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
    path = os.path.join(ROOT, "tests", "golden", "sfs_golden.json")
    with open(path, "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py (orc_ping_pong_bruteforce)", "cases": cases}, fh)
    print("wrote", path, sum(len(c["reads"]) for c in cases), "reads")


if __name__ == "__main__":
    main()
