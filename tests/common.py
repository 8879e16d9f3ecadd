"""Shared helpers for the tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT6 = {c: i for i, c in enumerate("$ACGTN")}


def from_ascii(s: str) -> np.ndarray:
    return np.array([NT6[c] for c in s], dtype=np.uint8)


def load_golden():
    with open(os.path.join(ROOT, "tests", "golden", "sfs_golden.json")) as fh:
        return json.load(fh)["cases"]


def split(counts, qs, ln):
    out, o = [], 0
    for c in counts.tolist():
        out.append(list(zip(qs[o:o + c].tolist(), ln[o:o + c].tolist())))
        o += c
    return out


def small_workload(seed=11, ref_lens=(150000, 80000), n_reads=64, read_len=1500, err=0.005, n_svs=6):
    """Seeded reference + haplotype with SVs + erroneous reads (HiFi shape, scaled down)."""
    from svdss_amd import synth
    ref = synth.make_reference(list(ref_lens), seed=seed, n_runs=(60,))
    hap, svs = synth.implant_svs(ref, n_svs, seed=seed + 1, min_len=50, max_len=400)
    flat, offs, truth = synth.simulate_reads(hap, n_reads, read_len, err, seed=seed + 2, ragged=True)
    return ref, hap, svs, flat, offs


# the SVDSS binary the process-level tests run (SVDSS_TEST_BIN: another build of it, e.g. the sanitized one of
# tests/test_sanitized_binary.py)
BIN = os.environ.get("SVDSS_TEST_BIN") or os.path.join(ROOT, "svdss_amd", "SVDSS")

