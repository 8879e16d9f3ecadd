"""Developer benchmark: `SVDSS index` -> `search` -> `call` end to end on a synthetic genome with implanted
SVs (error-free "smoothed" reads with truth alignments, as tests/pipeline_sim.py).  Runs on the GPU box.

  python tools/e2e_call.py [ref_bp] [n_svs] [coverage] [read_len] [workdir]
"""
import bisect
import json
import multiprocessing as mp
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svdss_amd import synth                      # noqa: E402
from tests.pipeline_sim import hap_segments     # noqa: E402
from tools.e2e_search import _bgzf_block, CODE16  # noqa: E402

OPS = "MIDNSHP=X"


def alignment(segs, starts, a, b):
    """tests/pipeline_sim.read_alignment restricted to the segments overlapping [a, b)."""
    i = max(0, bisect.bisect_right(starts, a) - 1)
    cigar, pos, last_r = [], None, None
    while i < len(segs) and segs[i][0] < b:
        hs, he, kind, rs = segs[i]
        lo, hi = max(a, hs), min(b, he)
        i += 1
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
            cigar.append(("S" if pos is None else "I", hi - lo))
    if cigar and cigar[-1][0] == "I":
        cigar[-1] = ("S", cigar[-1][1])
    merged = []
    for op, l in cigar:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + l)
        else:
            merged.append((op, l))
    return merged, pos


def write_dataset(work, ref_bp, n_svs, cov, read_len, seed=3, het_every=0, max_len=400):
    """FASTA + sorted error-free ("smoothed") BAM with truth alignments of reads drawn from a sample that carries
    n_svs implanted SVs.  het_every = k > 0: every k-th SV is heterozygous (half of the reads come from a second
    haplotype without it).  Returns (fa, bam, svs, het flags, [(pos, end, record bytes)], header bytes)."""
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(seed)
    ref = synth.make_reference([ref_bp], seed=seed, repeat_frac=0.0)
    hap, svs = synth.implant_svs(ref, n_svs, seed=seed + 1, min_len=60, max_len=max_len)
    fa = os.path.join(work, "ref.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n" + synth.to_ascii(ref[0]) + "\n")
    het = [het_every > 0 and (k % het_every) == het_every - 1 for k in range(len(svs))]
    haps = [(hap[0], [s for s in svs if s.contig == 0])]
    if any(het):
        # second haplotype: the homozygous SVs only (implant_svs is deterministic: rebuild from the kept ones)
        keep = [s for s, h in zip(svs, het) if not h]
        seq, last = [], 0
        for s in sorted(keep, key=lambda s: s.pos):
            seq.append(ref[0][last:s.pos])
            if s.kind == "INS":
                seq.append(s.seq)
                last = s.pos
            else:
                last = s.pos + s.length
        seq.append(ref[0][last:])
        haps.append((np.concatenate(seq), keep))
    lut = np.zeros(256, dtype=np.uint8)
    for ch, code in zip(b"ACGTN", (1, 2, 4, 8, 15)):
        lut[ch] = code
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrS\tLN:%d\n" % len(ref[0])
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1)
    hdr += struct.pack("<i", 5) + b"chrS\0" + struct.pack("<i", len(ref[0]))
    recs = []
    rid = 0
    for hseq, hsvs in haps:
        segs = hap_segments(len(ref[0]), hsvs)
        starts = [s[0] for s in segs]
        n = int(cov / len(haps) * len(hseq) / read_len)
        pos_a = np.sort(rng.integers(0, len(hseq) - read_len, size=n))
        hap_ascii = np.frombuffer(synth.to_ascii(hseq).encode(), dtype=np.uint8)
        for a in pos_a:
            cigar, pos = alignment(segs, starts, int(a), int(a) + read_len)
            rid += 1
            if pos is None:
                continue
            c = lut[hap_ascii[a:a + read_len]]
            if read_len & 1:
                c = np.append(c, np.uint8(0))
            packed = ((c[0::2] << 4) | c[1::2]).tobytes()
            name = ("read%07d" % rid).encode() + b"\0"
            cig = b"".join(struct.pack("<I", (l << 4) | OPS.index(op)) for op, l in cigar)
            core = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 60, 4680, len(cigar), 0, read_len, -1, -1, 0)
            body = core + name + cig + packed + b"\xff" * read_len + b"XFC\0"
            end = pos + sum(l for op, l in cigar if op in "MD")
            recs.append((pos, end, struct.pack("<i", len(body)) + body))
    recs.sort(key=lambda r: r[0])
    bam = os.path.join(work, "smoothed.bam")
    write_bam_records(bam, hdr, [r[2] for r in recs])
    return fa, bam, svs, het, recs, hdr, ref


def write_bam_records(path, hdr, records):
    data = hdr + b"".join(records)
    blocks = [data[i:i + 65280] for i in range(0, len(data), 65280)]
    with mp.Pool(min(64, os.cpu_count() or 1)) as pool:
        comp = pool.map(_bgzf_block, blocks, chunksize=64)
    with open(path, "wb") as f:
        for b in comp:
            f.write(b)
        f.write(_bgzf_block(b""))


def parse_vcf(vcf):
    called = []
    for line in vcf.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"])), f))
    return called


def main():
    ref_bp = int(sys.argv[1]) if len(sys.argv) > 1 else 20000000
    n_svs = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    cov = float(sys.argv[3]) if len(sys.argv) > 3 else 20
    read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
    work = sys.argv[5] if len(sys.argv) > 5 else "/tmp/svdss_e2e_call"
    t0 = time.time()
    fa, bam, svs, het, recs, hdr, ref = write_dataset(work, ref_bp, n_svs, cov, read_len)
    out = {"ref_bp": ref_bp, "n_svs": len(svs), "reads": len(recs), "read_len": read_len, "generate_s": round(time.time() - t0, 1)}
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    fmd = os.path.join(work, "ref.fmd")
    t0 = time.time()
    subprocess.run([exe, "index", "-t", str(os.cpu_count()), "-d", fa, "-o", fmd], check=True)
    out["index_s"] = round(time.time() - t0, 2)
    sfs = os.path.join(work, "specifics.txt")
    t0 = time.time()
    with open(sfs, "wb") as f:
        subprocess.run([exe, "search", "--index", fmd, "--bam", bam], check=True, stdout=f)
    out["search_s"] = round(time.time() - t0, 2)
    out["sfs_bytes"] = os.path.getsize(sfs)
    t0 = time.time()
    vcf = subprocess.run([exe, "call", "--reference", fa, "--bam", bam, "--sfs", sfs, "--threads", "16", "--min-sv-length", "50"] + sys.argv[6:],
                         check=True, stdout=subprocess.PIPE).stdout.decode()
    out["call_s"] = round(time.time() - t0, 2)
    called = [c[:3] for c in parse_vcf(vcf)]
    truth = [(s.pos, s.kind, s.length) for s in svs]
    hit = sum(1 for p, k, l in truth if any(k == ck and l == cl and abs(cp - p) <= 12 for cp, ck, cl in called))
    out.update({"called": len(called), "truth_recovered": hit})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
