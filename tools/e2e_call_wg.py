"""`SVDSS index` -> `search` -> `call` at the metric's scale: 24 contigs with the GRCh38 primary
lengths, ~3,400 implanted SVs (INS / DEL alternating, every other one heterozygous), >= 10^6 error-free ("smoothed")
15 kb reads with truth alignments (4.9x), a sorted BAM with its BAI, then the binaries with per-stage seconds of `call`
(run_svdss:142-178).  Runs on the GPU box; the generator uses every core it gets.

  python tools/e2e_call_wg.py [reads] [n_svs] [workdir] [scale]      scale < 1 shrinks every contig (trial runs)
  python tools/e2e_call_wg.py chain [reads] [n_svs] [workdir] [scale]
      the chain a user of run_svdss runs (run_svdss:136-178): the reads carry 0.5 % substitution errors and
      no XF tag; `SVDSS index` -> `SVDSS smooth` -> `SVDSS search` (putative: the reads smooth tagged XF != 0 are skipped) on the
      smoothed BAM -> `SVDSS call` on it; per-stage seconds, one search_plus_call_reads_per_s, truth recovery.
"""
import json
import multiprocessing as mp
import os
import re
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GRCH38_PRIMARY  # noqa: E402

L = 15000
CODE16 = np.array([0, 1, 2, 4, 8, 15], dtype=np.uint8)     # nt6 symbol -> BAM's 4-bit code
_G = {}


def _bgzf(data, level=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    cd = c.compress(data) + c.flush()
    return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cd) + 25) + cd + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return 4681 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return 585 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return 73 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return 9 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return 1 + (beg >> 26)
    return 0


def _contig(args):
    """one contig: reference, SVs, two haplotypes, reads, records, BGZF members.  Returns what the parent needs for the
    FASTA, the BAM and the BAI."""
    tid, ref_len, n_svs, n_reads, seed, work, err = args
    rng = np.random.default_rng(seed)
    ref = rng.integers(1, 5, size=ref_len, dtype=np.uint8)
    with open(os.path.join(work, "c%02d.fa" % tid), "wb") as f:
        f.write(b">c%d\n" % (tid + 1))
        f.write(np.frombuffer(b"$ACGTN", dtype=np.uint8)[ref].tobytes())
        f.write(b"\n")
    # SVs: one per stretch of ref_len / n_svs, away from the stretch's ends
    svs = []
    if n_svs:
        step = ref_len // n_svs
        for k in range(n_svs):
            ln = int(rng.integers(50, 2001))
            pos = k * step + int(rng.integers(step // 4, step // 2))
            kind = "INS" if k % 2 == 0 else "DEL"
            seq = rng.integers(1, 5, size=ln, dtype=np.uint8) if kind == "INS" else None
            svs.append((pos, kind, ln, seq, (k // 2) % 2 == 0))      # (.., heterozygous)
    haps = []
    for only_hom in (False, True):
        parts, segs, last, h = [], [], 0, 0
        for pos, kind, ln, seq, het in svs:
            if only_hom and het:
                continue
            if pos > last:
                segs.append((h, h + pos - last, "M", last))
                parts.append(ref[last:pos])
                h += pos - last
                last = pos
            if kind == "INS":
                segs.append((h, h + ln, "I", pos))
                parts.append(seq)
                h += ln
            else:
                last = pos + ln
        segs.append((h, h + ref_len - last, "M", last))
        parts.append(ref[last:])
        haps.append((CODE16[np.concatenate(parts)], segs))
    del ref
    recs = []   # (pos, end, bytes)
    qual = b"\xff" * L
    for hi, (code, segs) in enumerate(haps):
        n = n_reads // 2 + (n_reads % 2 if hi == 0 else 0)
        starts = np.sort(rng.integers(0, len(code) - L, size=n))
        seg_hs = np.array([s[0] for s in segs], dtype=np.int64)
        si = np.searchsorted(seg_hs, starts, side="right") - 1
        for r, (a, i) in enumerate(zip(starts.tolist(), si.tolist())):
            b = a + L
            hs, he, kind, rs = segs[i]
            if kind == "M" and b <= he:                        # the common read: inside one copied stretch
                cigar, pos, end = [(0, L)], rs + (a - hs), rs + (a - hs) + L
            else:
                cigar, pos, last_r = [], None, None
                j = i
                while j < len(segs) and segs[j][0] < b:
                    hs, he, kind, rs = segs[j]
                    lo, hi2 = max(a, hs), min(b, he)
                    j += 1
                    if lo >= hi2:
                        continue
                    if kind == "M":
                        r0 = rs + (lo - hs)
                        if pos is None:
                            pos = r0
                        elif r0 > last_r:
                            cigar.append((2, r0 - last_r))
                        cigar.append((0, hi2 - lo))
                        last_r = r0 + (hi2 - lo)
                    else:
                        cigar.append((4 if pos is None else 1, hi2 - lo))
                if cigar and cigar[-1][0] == 1:
                    cigar[-1] = (4, cigar[-1][1])
                if pos is None:
                    continue
                end = last_r
            c = code[a:b]
            if err > 0:                                      # substitution errors of a HiFi read (codes 1, 2, 4, 8 rotate)
                e = np.nonzero(rng.random(L) < err)[0]
                if len(e):
                    c = c.copy()
                    rot = rng.integers(1, 4, size=len(e))
                    idx = np.array([0, 0, 1, 0, 2, 0, 0, 0, 3], dtype=np.int64)[c[e]]
                    c[e] = np.array([1, 2, 4, 8], dtype=np.uint8)[(idx + rot) & 3]
            packed = ((c[0::2] << 4) | c[1::2]).tobytes()
            name = b"r%02d_%d_%07d\0" % (tid, hi, r)
            cig = b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar)
            core = struct.pack("<iiBBHHHiiii", tid, pos, len(name), 60, reg2bin(pos, end), len(cigar), 0, L, -1, -1, 0)
            body = core + name + cig + packed + qual + (b"XFC\0" if err <= 0 else b"")
            recs.append((pos, end, struct.pack("<i", len(body)) + body))
    recs.sort(key=lambda x: x[0])
    # BGZF members of this contig's records + where every record begins / ends (member, offset in member)
    members, cur, cur_len = [], [], 0
    voffs = []          # (member index, offset) of each record start; the end is the next record's start
    for pos, end, rec in recs:
        o = 0
        voffs.append((len(members), cur_len))
        while o < len(rec):
            take = min(len(rec) - o, 65280 - cur_len)
            cur.append(rec[o:o + take])
            cur_len += take
            o += take
            if cur_len == 65280:
                members.append(_bgzf(b"".join(cur)))
                cur, cur_len = [], 0
    voffs.append((len(members), cur_len))
    if cur_len:
        members.append(_bgzf(b"".join(cur)))
    path = os.path.join(work, "c%02d.bgzf" % tid)
    with open(path, "wb") as f:
        for m in members:
            f.write(m)
    sizes = np.array([len(m) for m in members], dtype=np.int64)
    return {"tid": tid, "n": len(recs), "pos": np.array([r[0] for r in recs], dtype=np.int64), "end": np.array([r[1] for r in recs], dtype=np.int64),
            "voff_member": np.array([v[0] for v in voffs], dtype=np.int64), "voff_off": np.array([v[1] for v in voffs], dtype=np.int64),
            "member_sizes": sizes, "svs": [(tid, p, k, ln, het) for p, k, ln, _, het in svs]}


def write_dataset(work, n_reads, n_svs, scale=1.0, err=0.0):
    os.makedirs(work, exist_ok=True)
    lens = [max(200000, int(x * scale)) for x in GRCH38_PRIMARY]
    total = sum(lens)
    jobs = []
    sv_left, rd_left = n_svs, n_reads
    for tid, ln in enumerate(lens):
        last = tid == len(lens) - 1
        ns = sv_left if last else round(n_svs * ln / total)
        nr = rd_left if last else round(n_reads * ln / total)
        sv_left -= ns
        rd_left -= nr
        jobs.append((tid, ln, ns, nr, 1000 + tid, work, err))
    # (largest contigs first; as many workers as memory allows: a worker holds ~4 bytes per base of its contig)
    with mp.Pool(min(12, os.cpu_count() or 1)) as pool:
        res = pool.map(_contig, sorted(jobs, key=lambda j: -j[1]), chunksize=1)
    res.sort(key=lambda r: r["tid"])
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:c%d\tLN:%d\n" % (i + 1, ln) for i, ln in enumerate(lens))
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(lens))
    for i, ln in enumerate(lens):
        nm = b"c%d\0" % (i + 1)
        hdr += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    fa, bam = os.path.join(work, "ref.fa"), os.path.join(work, "reads.bam")
    with open(fa, "wb") as f:
        for i in range(len(lens)):
            p = os.path.join(work, "c%02d.fa" % i)
            with open(p, "rb") as g:
                while True:
                    b = g.read(64 << 20)
                    if not b:
                        break
                    f.write(b)
            os.remove(p)
    bai = bytearray(b"BAI\1" + struct.pack("<i", len(lens)))
    with open(bam, "wb") as f:
        f.write(_bgzf(hdr))
        off = f.tell()
        for r in res:
            p = os.path.join(work, "c%02d.bgzf" % r["tid"])
            with open(p, "rb") as g:
                while True:
                    b = g.read(64 << 20)
                    if not b:
                        break
                    f.write(b)
            os.remove(p)
            # BAI of this contig (SAM specification 5.2): bins with chunks, 16 kb linear index
            mstart = off + np.concatenate([[0], np.cumsum(r["member_sizes"])])
            vm, vo = r["voff_member"], r["voff_off"]
            # "end of the previous member" form where a record ends exactly at a member boundary (bgzf_tell)
            def voff(k):
                m, o = int(vm[k]), int(vo[k])
                if o == 0 and m > 0 and k > 0:
                    return (int(mstart[m - 1]) << 16) | 65280
                return (int(mstart[m]) << 16) | o
            bins, lin = {}, {}
            for k in range(r["n"]):
                v0, v1 = voff(k), voff(k + 1)
                s, e = int(r["pos"][k]), int(r["end"][k])
                ch = bins.setdefault(reg2bin(s, e), [])
                if ch and ch[-1][1] == v0:
                    ch[-1][1] = v1
                else:
                    ch.append([v0, v1])
                for w in range(s >> 14, ((e - 1) >> 14) + 1):
                    lin.setdefault(w, v0)
            bai += struct.pack("<i", len(bins))
            for b in sorted(bins):
                bai += struct.pack("<Ii", b, len(bins[b]))
                for v0, v1 in bins[b]:
                    bai += struct.pack("<QQ", v0, v1)
            n_intv = max(lin) + 1 if lin else 0
            bai += struct.pack("<i", n_intv)
            last = 0
            for w in range(n_intv):
                last = lin.get(w, last)
                bai += struct.pack("<Q", last)
            off += int(r["member_sizes"].sum())
        f.write(_bgzf(b""))
    with open(bam + ".bai", "wb") as f:
        f.write(bytes(bai))
    svs = [s for r in res for s in r["svs"]]
    return fa, bam, svs, sum(r["n"] for r in res), lens


GEN_SRC = os.path.join(ROOT, "tools", "chain_dataset.cpp")
GEN_EXE = os.path.join(ROOT, "tools", "chain_dataset")


def build_generator():
    """tools/chain_dataset (C++; the Python generator above writes ~50 k reads/s with substitutions only)"""
    if not os.path.exists(GEN_EXE) or os.path.getmtime(GEN_EXE) < os.path.getmtime(GEN_SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", GEN_EXE, GEN_SRC, "-lz", "-ldl"], check=True)
    return GEN_EXE


def write_dataset_cxx(work, n_reads, n_svs, scale=1.0, err=0.005, threads=None, codec=None, keep_ref=False, ref_from_fasta=False):
    """The same data set by tools/chain_dataset.cpp, with errors sub:ins:del = 2:1.5:1.5 and the indels in the
    truth CIGARs.  Returns (fa, bam, svs, n, lens, info)."""
    os.makedirs(work, exist_ok=True)
    exe = build_generator()
    threads = threads or min(32, os.cpu_count() or 1)
    if codec is None:   # what htslib writes when it is built with libdeflate (level 6); zlib level 1 where that library is absent
        import ctypes.util
        codec = "libdeflate6" if ctypes.util.find_library("deflate") else "zlib1"
    r = subprocess.run([exe, work, str(n_reads), str(n_svs), repr(scale), repr(err), str(threads), codec, "2" if ref_from_fasta else "1" if keep_ref else "0"],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    svs = []
    with open(os.path.join(work, "truth.tsv")) as f:
        for line in f:
            t, p, k, ln, het = line.split()
            svs.append((int(t), int(p), k, int(ln), het == "1"))
    lens = [info["reference_bp"]] if ref_from_fasta else [max(200000, int(x * scale)) for x in GRCH38_PRIMARY]
    return os.path.join(work, "ref.fa"), os.path.join(work, "reads.bam"), svs, info["reads"], lens, info


def run(work, n_reads, n_svs, scale=1.0, threads=16):
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    out = {}
    t0 = time.perf_counter()
    fa, bam, svs, n, lens = write_dataset(work, n_reads, n_svs, scale)
    out["generate_s"] = round(time.perf_counter() - t0, 1)
    out.update({"reads": n, "svs": len(svs), "reference_bp": sum(lens), "bam_bytes": os.path.getsize(bam)})
    fmd = os.path.join(work, "ref.fmd")
    t0 = time.perf_counter()
    subprocess.run([exe, "index", "-d", fa, "-o", fmd], check=True, capture_output=True)
    out["index_s"] = round(time.perf_counter() - t0, 2)
    sfs = os.path.join(work, "specifics.txt")
    t0 = time.perf_counter()
    with open(sfs, "wb") as f:
        r = subprocess.run([exe, "search", "--index", fmd, "--bam", bam, "--verbose"], check=True, stdout=f, stderr=subprocess.PIPE, text=True)
    out["search_s"] = round(time.perf_counter() - t0, 3)
    m = re.search(r"on the device at \+([0-9.]+) s", r.stderr)
    out["search_index_resident_s"] = float(m.group(1)) if m else None
    out["sfs_bytes"] = os.path.getsize(sfs)
    t0 = time.perf_counter()
    c = subprocess.run([exe, "call", "--reference", fa, "--bam", bam, "--sfs", sfs, "--threads", str(threads), "--min-sv-length", "50", "--verbose"],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out["call_s"] = round(time.perf_counter() - t0, 3)
    out["call_log"] = [ln for ln in c.stderr.decode().splitlines() if "[time]" in ln or "debug" in ln][-40:]
    out["search_log"] = [ln for ln in r.stderr.splitlines() if "debug" in ln][-8:]
    called = []
    for line in c.stdout.decode().splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((f[0], int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"]))))
    truth = [("c%d" % (t + 1), p, k, ln) for t, p, k, ln, het in svs]
    by = {}
    for ch, p, k, ln in called:
        by.setdefault((ch, k, ln), []).append(p)
    hit = sum(1 for ch, p, k, ln in truth if any(abs(cp - p) <= 12 for cp in by.get((ch, k, ln), [])))
    out.update({"svs_called": len(called), "truth_recovered": hit, "call_reads_per_s": n / out["call_s"],
                "search_plus_call_reads_per_s": n / (out["search_s"] + out["call_s"])})
    return out


def _vcf_hits(vcf_text, svs):
    called = []
    for line in vcf_text.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((f[0], int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"]))))
    truth = [("c%d" % (t + 1), p, k, ln) for t, p, k, ln, het in svs]
    by = {}
    for ch, p, k, ln in called:
        by.setdefault((ch, k), []).append((p, ln))
    # (smoothed reads leave the SV's length exact; reads the accuracy filter left unsmoothed can move it by a base or two)
    hit = sum(1 for ch, p, k, ln in truth if any(abs(cp - p) <= 12 and abs(cl - ln) <= max(2, ln // 50) for cp, cl in by.get((ch, k), [])))
    return len(called), hit


def run_chain(work, n_reads, n_svs, scale=1.0, threads=16, err=0.005, generator="cxx", keep_ref=False, keep=False, stage_env=None,
              ref_from_fasta=False):
    """index -> smooth -> search (putative) -> call, as run_svdss:136-178 chains them, on reads WITH errors.
    generator "cxx": tools/chain_dataset.cpp, errors sub:ins:del = 2:1.5:1.5 with the indels in the CIGARs (round 6);
    "py": the Python generator above (substitutions only; rounds 4-5).  keep_ref: ref.fa / ref.fmd already in `work`
    (the same seeds give the same reference whatever the number of reads) are used as they are.  ref_from_fasta: the reads
    are drawn from the contigs of `work`/ref.fa (somebody else's reference: bench.py's), ref.fmd beside it is used if there."""
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    out = {}
    t0 = time.perf_counter()
    if generator == "cxx":
        fa, bam, svs, n, lens, info = write_dataset_cxx(work, n_reads, n_svs, scale, err=err, keep_ref=keep_ref, ref_from_fasta=ref_from_fasta)
        out["generator"] = "tools/chain_dataset.cpp: errors sub:ins:del = 2:1.5:1.5, %.1f CIGAR operations per read, BGZF by %s" % (info["cigar_ops_per_read"], info["codec"])
    else:
        fa, bam, svs, n, lens = write_dataset(work, n_reads, n_svs, scale, err=err)
        out["generator"] = "tools/e2e_call_wg.py: substitution errors only, BGZF by zlib level 1"
    out["generate_s"] = round(time.perf_counter() - t0, 1)
    out.update({"reads": n, "svs": len(svs), "read_error_rate": err, "reference_bp": sum(lens), "coverage": round(n * L / sum(lens), 2),
                "bam_bytes": os.path.getsize(bam)})
    fmd = os.path.join(work, "ref.fmd")
    env = dict(os.environ, **(stage_env or {}))
    if not ((keep_ref or ref_from_fasta) and os.path.exists(fmd)):
        t0 = time.perf_counter()
        subprocess.run([exe, "index", "-d", fa, "-o", fmd], check=True, capture_output=True, env=env)
        out["index_s"] = round(time.perf_counter() - t0, 2)
    sm = os.path.join(work, "smoothed.bam")
    t0 = time.perf_counter()
    with open(sm, "wb") as f:
        r = subprocess.run([exe, "smooth", "--reference", fa, "--bam", bam, "--threads", str(threads)], check=True, stdout=f, stderr=subprocess.PIPE,
                           text=True, env=dict(env, SVDSS_DEBUG="1"))
    out["smooth_s"] = round(time.perf_counter() - t0, 3)
    out["smooth_log"] = [ln for ln in r.stderr.splitlines() if "device path" in ln or "accuracy" in ln][-3:]
    out["smoothed_bam_bytes"] = os.path.getsize(sm)
    sfs = os.path.join(work, "specifics.txt")
    t0 = time.perf_counter()
    with open(sfs, "wb") as f:
        r = subprocess.run([exe, "search", "--index", fmd, "--bam", sm, "--verbose"], check=True, stdout=f, stderr=subprocess.PIPE, text=True, env=env)
    out["search_s"] = round(time.perf_counter() - t0, 3)
    m = re.search(r"on the device at \+([0-9.]+) s", r.stderr)
    out["search_index_resident_s"] = float(m.group(1)) if m else None
    out["sfs_bytes"] = os.path.getsize(sfs)
    out["search_log"] = [ln for ln in r.stderr.splitlines() if "debug" in ln][-6:]
    t0 = time.perf_counter()
    # (run_svdss:167-176 hands `call` the ORIGINAL BAM -- the one with an index beside it -- and the SFS of the smoothed reads)
    c = subprocess.run([exe, "call", "--reference", fa, "--bam", bam, "--sfs", sfs, "--threads", str(threads), "--min-sv-length", "50", "--verbose"],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    out["call_s"] = round(time.perf_counter() - t0, 3)
    out["call_bam"] = "the original BAM (its .bai beside it), as run_svdss does"
    out["call_log"] = [ln for ln in c.stderr.decode().splitlines() if "[time]" in ln or "pass 2 " in ln or "pass 1" in ln or "record store" in ln][-26:]
    n_called, hit = _vcf_hits(c.stdout.decode(), svs)
    chain = out["smooth_s"] + out["search_s"] + out["call_s"]
    out.update({"svs_called": n_called, "truth_recovered": hit, "smooth_reads_per_s": n / out["smooth_s"],
                "search_plus_call_s": round(out["search_s"] + out["call_s"], 3),
                "search_plus_call_reads_per_s": n / (out["search_s"] + out["call_s"]), "chain_reads_per_s": n / chain,
                "chain_s": round(chain, 3)})
    if keep:
        with open(os.path.join(work, "calls.vcf"), "wb") as f:
            f.write(c.stdout)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "chain":
        a = sys.argv[2:]
        print(json.dumps(run_chain(a[2] if len(a) > 2 else "/tmp/svdss_e2e_chain_wg", int(a[0]) if a else 1030000, int(a[1]) if len(a) > 1 else 3400,
                                   float(a[3]) if len(a) > 3 else 1.0, generator=os.environ.get("CHAIN_GENERATOR", "cxx"), keep=True,
                                   keep_ref=bool(os.environ.get("CHAIN_KEEP_REF"))), indent=1))
        sys.exit(0)
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1030000
    n_svs = int(sys.argv[2]) if len(sys.argv) > 2 else 3400
    work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/svdss_e2e_call_wg"
    scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    print(json.dumps(run(work, n_reads, n_svs, scale), indent=1))
