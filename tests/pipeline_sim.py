"""Error-free ("smoothed") read simulator with truth alignments for the call-stage tests: reads are
exact substrings of a haplotype carrying INS/DEL, their CIGARs follow from the haplotype->reference
map (what `SVDSS smooth` hands to `search`/`call`: reads equal the reference except at long indels)."""
import numpy as np

from svdss_amd import synth


def hap_segments(ref_len, svs):
    """[(hap_start, hap_end, kind, ref_start)] covering the haplotype: kind 'M' copies the reference,
    'I' is inserted sequence (ref_start = insertion point); deletions appear as jumps of ref_start."""
    segs, h, r = [], 0, 0
    for sv in sorted(svs, key=lambda s: s.pos):
        if sv.pos > r:
            segs.append((h, h + sv.pos - r, "M", r))
            h += sv.pos - r
            r = sv.pos
        if sv.kind == "INS":
            segs.append((h, h + sv.length, "I", r))
            h += sv.length
        else:
            r += sv.length
    if ref_len > r:
        segs.append((h, h + ref_len - r, "M", r))
    return segs


def read_alignment(segs, a, b):
    """CIGAR [(op, len)] and reference start of haplotype interval [a, b)."""
    cigar, pos, last_r = [], None, None
    for hs, he, kind, rs in segs:
        lo, hi = max(a, hs), min(b, he)
        if lo >= hi:
            continue
        if kind == "M":
            r0 = rs + (lo - hs)
            if pos is None:
                pos = r0
            elif last_r is not None and r0 > last_r:
                cigar.append(("D", r0 - last_r))
            cigar.append(("M", hi - lo))
            last_r = r0 + (hi - lo)
        else:
            if pos is None:                      # read starts inside an insertion: soft clip
                cigar.append(("S", hi - lo))
            else:
                cigar.append(("I", hi - lo))
    if cigar and cigar[-1][0] == "I":            # read ends inside an insertion: soft clip
        cigar[-1] = ("S", cigar[-1][1])
    merged = []
    for op, l in cigar:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + l)
        else:
            merged.append((op, l))
    return merged, pos


def simulate(ref_lens=(250000, 120000), n_svs=8, coverage=24, read_len=6000, seed=5, het_fraction=0.0):
    """Returns (ref contigs, svs, reads) with reads = [(name, tid, pos, cigar, seq_ascii, hp)]."""
    rng = np.random.default_rng(seed)
    ref = synth.make_reference(list(ref_lens), seed=seed, repeat_frac=0.0)
    hap, svs = synth.implant_svs(ref, n_svs, seed=seed + 1, min_len=60, max_len=400)
    reads = []
    k = 0
    for tid in range(len(ref)):
        tsvs = [s for s in svs if s.contig == tid]
        segs = hap_segments(len(ref[tid]), tsvs)
        ref_segs = hap_segments(len(ref[tid]), [])
        n = int(coverage * len(hap[tid]) / read_len)
        for _ in range(n):
            from_ref = rng.random() < het_fraction
            src, sg = (ref[tid], ref_segs) if from_ref else (hap[tid], segs)
            a = int(rng.integers(0, len(src) - read_len))
            cigar, pos = read_alignment(sg, a, a + read_len)
            if pos is None:
                continue
            reads.append((f"read{k:05d}", tid, pos, cigar, synth.to_ascii(src[a:a + read_len]), 0))
            k += 1
    reads.sort(key=lambda r: (r[1], r[2]))
    return ref, svs, reads


def add_errors(seq: str, cigar, rng, err: float):
    """HiFi-like errors on an error-free read, CIGAR kept truthful: inside M segments a base is
    substituted, followed by a 1-base insertion, or deleted (sub:ins:del = 2:1.5:1.5)."""
    out, new = [], []
    p = 0

    def push(op, l):
        if l <= 0:
            return
        if new and new[-1][0] == op:
            new[-1] = (op, new[-1][1] + l)
        else:
            new.append((op, l))

    for op, l in cigar:
        if op in ("I", "S"):
            out.append(seq[p:p + l])
            p += l
            push(op, l)
        elif op == "D":
            push(op, l)
        else:
            for i in range(l):
                b = seq[p + i]
                u = rng.random()
                first_or_last = i == 0 or i == l - 1     # keep segment borders clean (valid CIGAR shape)
                if u < err * 0.4:
                    out.append("ACGT"[("ACGT".index(b) + int(rng.integers(1, 4))) % 4])
                    push("M", 1)
                elif u < err * 0.7 and not first_or_last:
                    out.append(b)
                    out.append("ACGT"[int(rng.integers(0, 4))])
                    push("M", 1)
                    push("I", 1)
                elif u < err and not first_or_last:
                    push("D", 1)
                else:
                    out.append(b)
                    push("M", 1)
            p += l
    return "".join(out), new
