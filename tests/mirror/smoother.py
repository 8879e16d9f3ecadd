"""Host-side mirror of `SVDSS smooth` (/root/reference/smoother.cpp): rewrite every primary
alignment so that it equals the reference except at long (> 20 bp) indels and soft clips, and tag
it with XF.  A CIGAR walk with copies from the reference -- no DP.
"""
import math
from typing import Dict, List, Sequence, Tuple

from .clusterer import (Alignment, BAM_CDEL, BAM_CDIFF, BAM_CEQUAL, BAM_CINS, BAM_CMATCH, BAM_CSOFT_CLIP,
                        FLAG_SECONDARY, FLAG_SUPPLEMENTARY, FLAG_UNMAP)

MIN_INDEL_LENGTH = 20   # config.hpp:95, not settable


def _mismatch_counts(aln: Alignment, ref_seq: str) -> Tuple[int, int]:
    ref_off, q_off = aln.pos, 0
    n_match = n_mis = 0
    for l, op in aln.cigar:
        if op in (BAM_CMATCH, BAM_CEQUAL, BAM_CDIFF):
            for j in range(l):
                if ref_seq[ref_off + j] == aln.seq[q_off + j]:
                    n_match += 1
                else:
                    n_mis += 1
            ref_off += l
            q_off += l
        elif op == BAM_CINS or op == BAM_CSOFT_CLIP:
            q_off += l
        elif op == BAM_CDEL:
            ref_off += l
        else:
            break
    return n_match, n_mis


def eligible(aln: Alignment, ref_names, chromosomes, min_mapq) -> bool:
    """smoother.cpp:509-537: everything else is DROPPED from the output BAM."""
    if aln.flag & (FLAG_UNMAP | FLAG_SUPPLEMENTARY | FLAG_SECONDARY):
        return False
    if aln.mapq < min_mapq or len(aln.seq) < 2:
        return False
    return 0 <= aln.tid < len(ref_names) and ref_names[aln.tid] in chromosomes


def percentile(x: List[float], q: float) -> float:
    """smoother.cpp:246-255."""
    n = len(x)
    idx = (n - 1) * q
    lo, hi = math.floor(idx), math.ceil(idx)
    h = idx - lo
    return (1.0 - h) * x[lo] + h * x[hi]


def compute_maxaccuracy(alignments: Sequence[Alignment], ref_names, chromosomes, min_mapq=20, accp=0.98) -> float:
    """smoother.cpp:259-346: accp-percentile of mismatches/matches over the first 10 000 eligible alignments."""
    acc = []
    for a in alignments:
        if len(acc) >= 10000:
            break
        if not eligible(a, ref_names, chromosomes, min_mapq):
            continue
        m, x = _mismatch_counts(a, chromosomes[ref_names[a.tid]])
        acc.append(x / m if m else float("inf"))
    acc.sort()
    return percentile(acc, float(accp))


def smooth_read(aln: Alignment, qual: bytes, ref_seq: str, al_accuracy: float):
    """smoother.cpp:84-232.  Returns (xf, new_seq, new_qual, new_cigar); for xf != 0 the record keeps
    its original SEQ/QUAL/CIGAR and only gets the tag."""
    n_match = n_mis = 0
    new_seq: List[str] = []
    new_qual = bytearray()
    new_cigar: List[List[int]] = []
    ref_off, q_off, m_diff = aln.pos, 0, 0
    should_ignore = True

    def qslice(start, ln):   # the reference copies `ln` quality bytes from `start` (may run past the end)
        s = qual[start:start + ln]
        return s + b"\xff" * (ln - len(s))

    for l, op in aln.cigar:
        if op in (BAM_CMATCH, BAM_CEQUAL, BAM_CDIFF):
            new_seq.append(ref_seq[ref_off:ref_off + l])
            new_qual += qslice(q_off, l)
            for j in range(l):
                if ref_seq[ref_off + j] == aln.seq[q_off + j]:
                    n_match += 1
                else:
                    n_mis += 1
            ref_off += l
            q_off += l
            if new_cigar and new_cigar[-1][1] == BAM_CMATCH:
                new_cigar[-1][0] += l + m_diff
            else:
                new_cigar.append([l + m_diff, BAM_CMATCH])
            m_diff = 0
        elif op == BAM_CINS:
            if l > MIN_INDEL_LENGTH:
                should_ignore = False
                new_seq.append(aln.seq[q_off:q_off + l])
                new_qual += qslice(q_off, l)
                new_cigar.append([l, op])
            q_off += l
        elif op == BAM_CDEL:
            if l <= MIN_INDEL_LENGTH:
                new_seq.append(ref_seq[ref_off:ref_off + l])
                new_qual += qslice(q_off, l)       # qualities of the NEXT read bases (App. A#16)
                m_diff += l
            else:
                should_ignore = False
                new_cigar.append([l, op])
            ref_off += l
        elif op == BAM_CSOFT_CLIP:
            should_ignore = False
            new_seq.append(aln.seq[q_off:q_off + l])
            new_qual += qslice(q_off, l)
            q_off += l
            new_cigar.append([l, op])
        else:
            break
    ratio = (n_mis / n_match) if n_match else float("inf")
    if ratio > al_accuracy:
        return 1, None, None, None
    if should_ignore:
        return 2, None, None, None
    return 0, "".join(new_seq), bytes(new_qual), [(l, op) for l, op in new_cigar]
