"""Seeded synthetic data in HiFi shape: references with
diverged repeats, a sample haplotype carrying SVs, reads with sub/ins/del errors.

Everything is nt6-coded uint8 (A=1 C=2 G=3 T=4 N=5).  This is data plumbing for
tests and bench.py, not part of the hot path.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

COMP = np.array([0, 4, 3, 2, 1, 5], dtype=np.uint8)


def revcomp(a: np.ndarray) -> np.ndarray:
    return COMP[a[::-1]]


def make_reference(lengths, seed: int, repeat_frac: float = 0.03, divergence: float = 0.01,
                   n_runs: Tuple[int, ...] = ()) -> List[np.ndarray]:
    """iid ACGT contigs; repeat_frac of each contig is overwritten by copies of
    earlier segments (1-10 kb, random strand) mutated at `divergence`; optional N runs."""
    rng = np.random.default_rng(seed)
    contigs = []
    for L in lengths:
        c = rng.integers(1, 5, size=L, dtype=np.uint8)
        target = int(L * repeat_frac)
        done = 0
        while done < target and L > 4000:
            seg = int(rng.integers(1000, min(10000, L // 4) + 1))
            src = int(rng.integers(0, L - seg))
            dst = int(rng.integers(0, L - seg))
            piece = c[src:src + seg].copy()
            if rng.random() < 0.5:
                piece = revcomp(piece)
            mut = rng.random(seg) < divergence
            piece[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
            c[dst:dst + seg] = piece
            done += seg
        for run in n_runs:
            if run < L:
                s = int(rng.integers(0, L - run))
                c[s:s + run] = 5
        contigs.append(c)
    return contigs


def make_family_reference(lengths, seed: int, repeat_frac: float = 0.45, divergence: float = 0.10,
                          n_families: int = 40) -> List[np.ndarray]:
    """iid ACGT contigs in which repeat_frac of the bases are copies of a few repeat families (consensus of 300 bp --
    6 kb, every copy on a random strand, mutated at `divergence`, truncated at a random point like 5'-truncated L1
    copies): thousands of copies per family, the way half of a mammalian genome looks, instead of make_reference's
    pairwise segmental copies."""
    rng = np.random.default_rng(seed)
    fam = [rng.integers(1, 5, size=int(n), dtype=np.uint8)
           for n in rng.choice([300, 300, 300, 1000, 2000, 6000], size=n_families)]
    contigs = []
    for L in lengths:
        c = rng.integers(1, 5, size=L, dtype=np.uint8)
        target, done = int(L * repeat_frac), 0
        while done < target and L > 8000:
            f = fam[int(rng.integers(0, n_families))]
            n = len(f) if rng.random() < 0.5 else int(rng.integers(100, len(f) + 1))
            piece = f[len(f) - n:].copy()
            mut = rng.random(n) < divergence
            piece[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
            if rng.random() < 0.5:
                piece = revcomp(piece)
            dst = int(rng.integers(0, L - n))
            c[dst:dst + n] = piece
            done += n
        contigs.append(c)
    return contigs


@dataclass
class SV:
    contig: int
    pos: int        # 0-based reference position (first affected base / insertion point)
    kind: str       # "INS" or "DEL"
    length: int
    seq: np.ndarray = field(default=None, repr=False)


def implant_svs(contigs: List[np.ndarray], n_svs: int, seed: int, min_len: int = 50,
                max_len: int = 2000, window: Tuple[int, int, int] = None):
    """Sample haplotype = reference with n_svs non-overlapping INS/DEL.
    window = (contig, start, end) restricts placement.  Returns (haplotype contigs, [SV])."""
    rng = np.random.default_rng(seed)
    svs: List[SV] = []
    per_contig = {i: [] for i in range(len(contigs))}
    tries = 0
    while len(svs) < n_svs and tries < 100 * n_svs + 100:
        tries += 1
        if window is not None:
            ci, ws, we = window
        else:
            ci = int(rng.integers(0, len(contigs)))
            ws, we = 0, len(contigs[ci])
        ln = int(rng.integers(min_len, max_len + 1))
        if we - ws <= 2 * ln + 2000:
            continue
        pos = int(rng.integers(ws + 1000, we - ln - 1000))
        if any(abs(pos - s.pos) < 3 * max_len for s in per_contig[ci]):
            continue
        kind = "INS" if len(svs) % 2 == 0 else "DEL"
        seq = rng.integers(1, 5, size=ln, dtype=np.uint8) if kind == "INS" else None
        sv = SV(ci, pos, kind, ln, seq)
        svs.append(sv)
        per_contig[ci].append(sv)
    hap = []
    for ci, c in enumerate(contigs):
        parts, prev = [], 0
        for sv in sorted(per_contig[ci], key=lambda s: s.pos):
            parts.append(c[prev:sv.pos])
            if sv.kind == "INS":
                parts.append(sv.seq)
                prev = sv.pos
            else:
                prev = sv.pos + sv.length
        parts.append(c[prev:])
        hap.append(np.concatenate(parts))
    return hap, svs


def simulate_reads(contigs: List[np.ndarray], n_reads: int, read_len: int, err: float, seed: int,
                   window: Tuple[int, int, int] = None, ragged: bool = False):
    """Reads of `read_len` bases (or U[read_len/2, read_len] if ragged) from random
    positions and strands of `contigs`, errors at rate `err` split sub:ins:del = 2:1.5:1.5.
    Returns (flat uint8 reads, int64 offsets[n+1], truth list of (contig, start, strand))."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    reads, truth = [], []
    p_sub, p_ins, p_del = err * 0.4, err * 0.3, err * 0.3
    for _ in range(n_reads):
        L = int(rng.integers(read_len // 2, read_len + 1)) if ragged else read_len
        if window is not None:
            ci, ws, we = window
        else:
            ci = int(rng.choice(len(contigs), p=lens / lens.sum()))
            ws, we = 0, int(lens[ci])
        span = L + L // 20 + 64
        span = min(span, we - ws)
        start = int(rng.integers(ws, we - span + 1))
        src = contigs[ci][start:start + span]
        u = rng.random(span)
        kind = np.zeros(span, dtype=np.int8)           # 0 match, 1 sub, 2 ins (after base), 3 del
        kind[u < p_sub] = 1
        kind[(u >= p_sub) & (u < p_sub + p_ins)] = 2
        kind[(u >= p_sub + p_ins) & (u < p_sub + p_ins + p_del)] = 3
        reps = np.ones(span, dtype=np.int64)
        reps[kind == 2] = 2
        reps[kind == 3] = 0
        out = np.repeat(src, reps)
        idx = np.repeat(np.arange(span), reps)
        first = np.ones(len(out), dtype=bool)
        first[1:] = idx[1:] != idx[:-1]
        subs = (kind[idx] == 1) & first
        # substitution: a different base (only meaningful for ACGT)
        shift = rng.integers(1, 4, size=len(out), dtype=np.uint8)
        acgt = (out >= 1) & (out <= 4)
        out = np.where(subs & acgt, ((out - 1 + shift) % 4) + 1, out).astype(np.uint8)
        inserted = ~first
        out[inserted] = rng.integers(1, 5, size=int(inserted.sum()), dtype=np.uint8)
        out = out[:L]
        strand = int(rng.random() < 0.5)
        if strand:
            out = revcomp(out)
        reads.append(np.ascontiguousarray(out))
        truth.append((ci, start, strand))
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(r) for r in reads])
    flat = np.concatenate(reads) if reads else np.zeros(0, dtype=np.uint8)
    return flat, offsets, truth


def to_ascii(a: np.ndarray) -> str:
    return bytes(np.frombuffer(b"$ACGTN", dtype=np.uint8)[a]).decode()
