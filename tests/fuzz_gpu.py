"""Time-boxed differential fuzzing of the HIP kernels against their checkers (test infrastructure, like everything
under tests/: the oracle is the checker, never the thing measured).

    python -m tests.fuzz_gpu --what search,poa,extd2,ratio,inflate,deflate,place,index --minutes 5 --seed 1

Each fuzzer draws adversarial inputs the fixed-seed tests of this directory do not reach by construction -- references
that are tandem repeats / homopolymers / palindromes / one base long, reads that span contig junctions, clusters of
reads of wildly different lengths, damaged deflate streams -- runs the kernel through the C-ABI and compares bit for
bit with oracle/ (search, POA, extd2, ratio), zlib (inflate, deflate), the host builder (index) or the Python
restatement (placement).  The first mismatch is written to --out as an .npz / .bin with its seed and the process
exits 1; a summary line per fuzzer otherwise.  Runs on a GPU box (`gpurun`), minutes at a time; the summaries of the
runs made are under profiles/."""
import argparse
import os
import sys
import time
import zlib

import numpy as np

import svdss_amd
from svdss_amd import synth
from tests import oracle_lib as O

LET = np.frombuffer(b"ACGTN", dtype=np.uint8)


class Mismatch(Exception):
    pass


def _dump(out_dir, name, **items):
    """the failing case: arrays as they are, lists of arrays as flat + offsets, everything else as text"""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name + ".npz")
    save = {}
    for k, v in items.items():
        if isinstance(v, list):
            arrs = [np.asarray(a) for a in v]
            save[k + "_flat"] = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
            save[k + "_off"] = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.int64)
        elif isinstance(v, np.ndarray):
            save[k] = v
        else:
            save[k] = np.frombuffer(str(v).encode(), np.uint8)
    np.savez_compressed(path, **save)
    return path


# ------------------------------------------------------------------------------------------------ search

def _weird_contig(rng, max_len):
    kind = int(rng.integers(0, 9))
    n = int(rng.integers(1, max_len + 1))
    if kind == 0:                                         # iid
        c = rng.integers(1, 5, size=n, dtype=np.uint8)
    elif kind == 1:                                       # two letters
        c = rng.choice(np.array([1, 3], np.uint8), size=n)
    elif kind == 2:                                       # homopolymer runs
        runs = rng.geometric(0.08, size=n // 4 + 2)
        c = np.repeat(rng.integers(1, 5, size=len(runs), dtype=np.uint8), runs)[:n]
    elif kind == 3:                                       # a tandem repeat, lightly mutated
        unit = rng.integers(1, 5, size=int(rng.integers(1, 61)), dtype=np.uint8)
        c = np.tile(unit, n // len(unit) + 1)[:n].copy()
        m = rng.random(n) < float(rng.choice([0, 0.001, 0.02]))
        c[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
    elif kind == 4:                                       # families of near-identical copies
        n = max(n, 9000)
        c = synth.make_family_reference([n], seed=int(rng.integers(1 << 30)), repeat_frac=0.7,
                                        divergence=float(rng.choice([0.0, 0.002, 0.02])), n_families=3)[0]
    elif kind == 5:                                       # a palindrome: the sequence and its reverse complement
        h = rng.integers(1, 5, size=max(1, n // 2), dtype=np.uint8)
        c = np.concatenate([h, synth.revcomp(h)])
    elif kind == 6:                                       # very short
        c = rng.integers(1, 5, size=int(rng.integers(1, 40)), dtype=np.uint8)
    elif kind == 7:                                       # one symbol
        c = np.full(min(n, 3000), int(rng.integers(1, 5)), np.uint8)
    else:                                                 # iid with N runs, also at the ends
        c = rng.integers(1, 5, size=n, dtype=np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            l = int(rng.integers(1, max(2, min(300, n))))
            s = int(rng.choice([0, max(0, n - l), int(rng.integers(0, max(1, n - l + 1)))]))
            c[s:s + l] = 5
    return np.ascontiguousarray(c, dtype=np.uint8)


def _weird_reads(rng, contigs, n_reads):
    reads = []
    for _ in range(n_reads):
        kind = int(rng.integers(0, 10))
        ci = int(rng.integers(0, len(contigs)))
        c = contigs[ci]
        if kind <= 3:                                     # a stretch of a contig with errors, either strand
            L = int(rng.integers(1, min(len(c), 4000) + 1))
            s = int(rng.integers(0, len(c) - L + 1))
            r = c[s:s + L].copy()
            e = float(rng.choice([0, 0, 0.001, 0.01, 0.05, 0.2]))
            m = rng.random(L) < e
            r[m] = rng.integers(1, 5, size=int(m.sum()), dtype=np.uint8)
            if e and rng.random() < 0.5 and L > 10:       # an indel
                at = int(rng.integers(1, L - 1))
                r = np.concatenate([r[:at], rng.integers(1, 5, size=int(rng.integers(1, 30)), dtype=np.uint8), r[at:]]) \
                    if rng.random() < 0.5 else np.concatenate([r[:at], r[at + int(rng.integers(1, min(30, L - at))):]])
            if rng.random() < 0.5:
                r = synth.revcomp(r)
        elif kind == 4:                                   # the end of one contig and the start of another (or of its reverse complement)
            d = contigs[int(rng.integers(0, len(contigs)))]
            a = c[max(0, len(c) - int(rng.integers(1, 200))):]
            b = d[:int(rng.integers(1, 200))]
            r = np.concatenate([a, synth.revcomp(b) if rng.random() < 0.3 else b])
        elif kind == 5:                                   # unrelated sequence
            r = rng.integers(1, 5, size=int(rng.integers(1, 1500)), dtype=np.uint8)
        elif kind == 6:                                   # a whole contig and a little more
            r = np.concatenate([c[:5000], rng.integers(1, 5, size=int(rng.integers(0, 5)), dtype=np.uint8)])
        elif kind == 7:                                   # with N
            L = int(rng.integers(1, min(len(c), 2000) + 1))
            s = int(rng.integers(0, len(c) - L + 1))
            r = c[s:s + L].copy()
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, L))
                r[at:at + int(rng.integers(1, 6))] = 5
        elif kind == 8:                                   # a homopolymer / short tandem repeat
            unit = rng.integers(1, 5, size=int(rng.integers(1, 7)), dtype=np.uint8)
            r = np.tile(unit, int(rng.integers(1, 300)))
        else:                                             # empty or one base
            r = rng.integers(1, 6, size=int(rng.integers(0, 2)), dtype=np.uint8)
        reads.append(np.ascontiguousarray(r, dtype=np.uint8))
    return reads


def fuzz_search(rng, out_dir, it):
    n_contigs = int(rng.integers(1, 5))
    contigs = [_weird_contig(rng, int(rng.choice([60, 3000, 40000, 150000]))) for _ in range(n_contigs)]
    env = {}
    k = rng.choice(["", "4", "7", "10", "12", "13"])
    if k:
        env["SVDSS_KMER"] = str(k)
    on_device = rng.random() < 0.6
    if on_device and rng.random() < 0.4:
        env["SVDSS_SA_PIECE"] = str(int(rng.choice([64, 1000, 20000])))
    if rng.random() < 0.25:
        env["SVDSS_FORCE_SA64"] = "1"
    if rng.random() < 0.3:
        env["SVDSS_SEGMENTS"] = "1"
    reads = _weird_reads(rng, contigs, int(rng.integers(1, 500)))
    flat, offs = svdss_amd.pack_reads(reads)
    old = {k_: os.environ.get(k_) for k_ in ("SVDSS_KMER", "SVDSS_SA_PIECE", "SVDSS_FORCE_SA64", "SVDSS_SEGMENTS")}
    for k_ in old:
        os.environ.pop(k_, None)
    os.environ.update(env)
    try:
        ix = svdss_amd.FMDIndex.build(contigs, device=0) if on_device else svdss_amd.FMDIndex.build(contigs, threads=2).to_device(0)
        fm = O.OracleFMD.build(contigs)
        if not (ix.bwt() == fm.bwt()).all():
            raise Mismatch("BWT differs: " + _dump(out_dir, f"search_{it}", contigs=contigs, env=str(env)))
        v = ix.verify(1)
        if v["bad_order"] or v["bad_bwt"] or v["bad_range"] or v["bad_block"] or v["bad_dollar"]:
            raise Mismatch(f"index verify {v}: " + _dump(out_dir, f"search_{it}", contigs=contigs, env=str(env)))
        for assemble in (False, True):
            pp = svdss_amd.PingPong(ix, assemble=assemble)
            got = pp.ping_pong_search(flat, offs)
            pp.close()
            c, q, l, e = fm.search_batch(flat, offs, assemble)
            if not ((got.counts == c).all() and (got.n_ext == e).all() and len(got.qs) == len(q) and (got.qs == q).all() and (got.len == l).all()):
                bad = np.nonzero((got.counts != c) | (got.n_ext != e))[0]
                raise Mismatch(f"search differs (assemble={assemble}, env={env}, first bad read {bad[:1]}): " +
                               _dump(out_dir, f"search_{it}", contigs=contigs, reads=reads, env=str(env)))
        ix.close()
    finally:
        for k_ in env:
            os.environ.pop(k_, None)
        for k_, v_ in old.items():
            if v_ is not None:
                os.environ[k_] = v_
    return int(offs[-1]), int(c.sum())


# ------------------------------------------------------------------------------------------------ POA

def _mutate(rng, s, e):
    out = []
    for b in s.tolist():
        x = rng.random()
        if x < e * 0.4:
            out.append(int(rng.integers(0, 4)))
        elif x < e * 0.7:
            out.append(b); out.append(int(rng.integers(0, 4)))
        elif x < e:
            continue
        else:
            out.append(b)
    return np.array(out, np.uint8)


def _weird_cluster(rng):
    kind = int(rng.integers(0, 8))
    L = int(rng.choice([5, 40, 200, 200, 700, 700, 1500, 2400, 5600 if rng.random() < 0.15 else 300]))
    if kind == 0:
        t = rng.integers(0, 4, size=L, dtype=np.uint8)
    elif kind == 1:                                       # tandem repeat: every alignment has ties
        u = rng.integers(0, 4, size=int(rng.integers(1, 9)), dtype=np.uint8)
        t = np.tile(u, L // len(u) + 1)[:L]
    elif kind == 2:                                       # homopolymer runs
        runs = rng.geometric(0.25, size=L)
        t = np.repeat(rng.integers(0, 4, size=L, dtype=np.uint8), runs)[:L]
    else:
        t = rng.integers(0, 4, size=L, dtype=np.uint8)
    n = int(rng.choice([1, 2, 3, 5, 8, 12, 20, 45]))
    if L > 3000:
        n = min(n, 4)
    e = float(rng.choice([0, 0.005, 0.02, 0.1, 0.2]))
    alts = [t]
    for _ in range(int(rng.integers(0, 4))):              # haplotypes: an insertion, a deletion, a duplication, a different start
        a = t
        at = int(rng.integers(0, len(a) + 1))
        x = int(rng.integers(0, 4))
        ln = int(rng.integers(1, max(2, min(600, len(a)))))
        if x == 0:
            a = np.concatenate([a[:at], rng.integers(0, 4, size=ln, dtype=np.uint8), a[at:]])
        elif x == 1:
            a = np.concatenate([a[:at], a[at + ln:]])
        elif x == 2:
            a = np.concatenate([a[:at], a[max(0, at - ln):at], a[at:]])
        else:
            a = a[min(len(a) - 1, ln):] if rng.random() < 0.5 else a[:max(1, len(a) - ln)]
        alts.append(np.ascontiguousarray(a, dtype=np.uint8))
    reads = []
    for i in range(n):
        r = _mutate(rng, alts[int(rng.integers(0, len(alts)))], e)
        x = rng.random()
        if x < 0.03:
            r = np.zeros(0, np.uint8)
        elif x < 0.06:
            r = rng.integers(0, 4, size=int(rng.integers(1, 300)), dtype=np.uint8)       # unrelated
        elif x < 0.10 and len(r):
            r = r.copy(); r[int(rng.integers(0, len(r)))] = 4                               # an N
        elif x < 0.12:
            r = np.full(int(rng.integers(1, 50)), 4, np.uint8)
        reads.append(np.ascontiguousarray(r, dtype=np.uint8))
    if kind in (3, 4):                                    # lengths all over the place: slices of the template, unrelated reads, runs
        reads = []                                        # (bands that stand still, jump, lose the sink: fuzz seed 7 / 105)
        for i in range(min(n, 12) + 1):
            x = rng.random()
            if x < 0.4:
                a = int(rng.integers(0, len(t)))
                r = _mutate(rng, t[a:a + int(rng.integers(1, len(t) - a + 1))], e)
            elif x < 0.7:
                runs = rng.geometric(float(rng.choice([0.15, 0.3, 0.6])), size=len(t) + 1)
                r = np.repeat(rng.integers(0, 4, size=len(t) + 1, dtype=np.uint8), runs)[:int(rng.integers(1, len(t) + 1))]
            elif x < 0.85:
                r = rng.integers(0, 4, size=int(rng.integers(1, 2 * len(t) + 1)), dtype=np.uint8)
            else:
                r = _mutate(rng, t, e)
            reads.append(np.ascontiguousarray(r, dtype=np.uint8))
        if L > 3000:
            reads = reads[:4]
    if kind == 7:                                         # many different insertions at one place: a node with many predecessors
        at = len(t) // 2
        reads = [t] + [np.concatenate([t[:at], rng.integers(0, 4, size=3 + i, dtype=np.uint8), t[at:]]) for i in range(int(rng.integers(2, 14)))]
    return reads


def fuzz_poa(rng, out_dir, it):
    from svdss_amd import calldp
    clusters = [_weird_cluster(rng) for _ in range(int(rng.integers(4, 40)))]
    got, stats = calldp.run_poa(clusters)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:     # (the oracle call releases the GIL)
        wants = list(ex.map(lambda reads: bytes(LET[O.poa_consensus(reads)]).decode(), clusters))
    for k, (reads, g, want) in enumerate(zip(clusters, got, wants)):
        if g != want:
            raise Mismatch(f"POA consensus differs (cluster {k}, {len(reads)} reads): " +
                           _dump(out_dir, f"poa_{it}", reads=[np.asarray(r) for r in reads], got=g, want=want))
    return int(stats["cells"]), len(clusters)


# ------------------------------------------------------------------------------------------------ extd2 / ratio

def fuzz_extd2(rng, out_dir, it):
    from svdss_amd import calldp
    qs, ts = [], []
    for _ in range(int(rng.integers(8, 120))):
        kind = int(rng.integers(0, 5))
        tl = int(rng.choice([0, 1, 3, 50, 300, 1200, 2500]))
        tl = int(rng.integers(0, tl + 1))
        if kind == 1:
            u = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
            t = np.tile(u, tl // len(u) + 1)[:tl]
        elif kind == 2:
            t = np.repeat(rng.integers(0, 4, size=tl + 1, dtype=np.uint8), rng.geometric(0.3, size=tl + 1))[:tl]
        else:
            t = rng.integers(0, 5 if kind == 3 else 4, size=tl, dtype=np.uint8)
        q = _mutate(rng, t, float(rng.choice([0, 0.01, 0.1, 0.3])))
        if len(q) > 4 and rng.random() < 0.6:
            at = int(rng.integers(0, len(q)))
            ln = int(rng.integers(1, 500))
            q = np.concatenate([q[:at], rng.integers(0, 4, size=ln, dtype=np.uint8), q[at:]]) if rng.random() < 0.5 else np.concatenate([q[:at], q[at + ln:]])
        if rng.random() < 0.05:
            q = rng.integers(0, 4, size=int(rng.integers(0, 400)), dtype=np.uint8)
        qs.append(np.ascontiguousarray(q, dtype=np.uint8)); ts.append(np.ascontiguousarray(t, dtype=np.uint8))
    scores, cigars, stats = calldp.ksw_extd2_global(qs, ts)
    for k, (q, t, s, c) in enumerate(zip(qs, ts, scores.tolist(), cigars)):
        es, ec = O.ksw_extd2_global(q, t, calldp.KSW_MAT)
        if s != es or c.tolist() != ec.tolist():
            raise Mismatch(f"extd2 differs (pair {k}: score {s} vs {es}): " + _dump(out_dir, f"extd2_{it}", q=q, t=t, got=np.asarray(c), want=np.asarray(ec)))
    return int(stats["cells"]), len(qs)


def fuzz_ratio(rng, out_dir, it):
    from svdss_amd import calldp
    a_list, b_list = [], []
    alpha = int(rng.choice([1, 2, 4, 5, 8, 9, 30, 256]))
    syms = rng.choice(256, size=alpha, replace=False).astype(np.uint8)
    for _ in range(int(rng.integers(10, 200))):
        la = int(rng.integers(0, int(rng.choice([2, 70, 130, 700, 4200, 7000])) + 1))
        a = rng.choice(syms, size=la)
        x = int(rng.integers(0, 4))
        if x == 0:
            b = a.copy()
        elif x == 1:
            b = rng.choice(syms, size=int(rng.integers(0, 3000)))
        else:
            b = a.copy()
            for _ in range(int(rng.integers(0, 40))):
                if len(b):
                    b[int(rng.integers(0, len(b)))] = rng.choice(syms)
            if len(b) > 3:
                at = int(rng.integers(0, len(b)))
                b = np.delete(b, slice(at, at + int(rng.integers(0, 300)))) if x == 2 else np.insert(b, at, rng.choice(syms, size=int(rng.integers(0, 300))))
        a_list.append(bytes(a.astype(np.uint8))); b_list.append(bytes(b.astype(np.uint8)))
    if rng.random() < 0.3:
        os.environ["SVDSS_RATIO_DP"] = "1"
    try:
        ratio, lcs = calldp.fuzz_ratio(a_list, b_list)
    finally:
        os.environ.pop("SVDSS_RATIO_DP", None)
    for k, (a, b, r, l) in enumerate(zip(a_list, b_list, ratio.tolist(), lcs.tolist())):
        if l != O.lcs(a, b) or r != O.fuzz_ratio(a, b):
            raise Mismatch(f"ratio differs (pair {k}, {len(a)} x {len(b)}, alphabet {alpha}): " +
                           _dump(out_dir, f"ratio_{it}", a=np.frombuffer(a, np.uint8), b=np.frombuffer(b, np.uint8)))
    return sum(len(a) * len(b) for a, b in zip(a_list, b_list)), len(a_list)


# ------------------------------------------------------------------------------------------------ inflate / deflate

def _mixture(rng, target):
    buf = bytearray()
    while len(buf) < target:
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(1, 4000))
        if kind == 0:
            a = int(rng.choice([1, 2, 3, 4, 16, 40, 94, 200, 256]))
            buf += rng.integers(0, a, size=n, dtype=np.uint8).tobytes()
        elif kind == 1:
            buf += bytes([int(rng.integers(0, 256))]) * n
        elif kind == 2 and buf:
            d = int(rng.integers(1, min(len(buf), 32768) + 1))
            for _ in range(n):
                buf.append(buf[-d])
        elif kind == 3:
            w = bytes(rng.integers(97, 123, size=int(rng.integers(2, 12)), dtype=np.uint8))
            buf += (w + b" ") * (n // len(w) + 1)
        elif kind == 4:
            p_ = np.r_[float(rng.choice([0.5, 0.9, 0.99])), np.zeros(254)]
            p_[1:] = (1 - p_[0]) / 254
            buf += rng.choice(np.arange(255, dtype=np.uint8), p=p_, size=n).tobytes()
        else:                                             # geometric: code lengths up to the limit
            buf += np.minimum(255, rng.geometric(float(rng.choice([0.5, 0.1, 0.02])), size=n)).astype(np.uint8).tobytes()
    return bytes(buf[:target])


def _zlib_inflate(raw, isize):
    """(ok, bytes): what zlib makes of a raw deflate stream that should hold isize bytes"""
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(raw, isize + 1)
        if len(out) <= isize and not d.eof:
            out += d.flush()
    except zlib.error:
        return False, b""
    return (d.eof and len(out) == isize), out


def fuzz_inflate(rng, out_dir, it):
    from svdss_amd._lib import SvdssError
    from svdss_amd.bgzf import gpu_inflate
    streams, want = [], []
    for _ in range(int(rng.integers(20, 200))):
        data = _mixture(rng, int(rng.integers(0, 65537)))
        level = int(rng.choice([0, 1, 1, 2, 4, 6, 9]))
        strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 4, 8, 9])), strat)
        s = c.compress(data)
        if rng.random() < 0.2:                            # several deflate blocks, some of them empty stored ones
            s += c.flush(zlib.Z_FULL_FLUSH) + c.compress(b"")
        s += c.flush()
        streams.append(s); want.append(data)
    comp, blocks, uoff, at = bytearray(), [], [], 64
    for s_, w in zip(streams, want):
        comp += b"\x5a" * int(rng.integers(0, 5))
        blocks.append((len(comp), len(s_), len(w)))
        comp += s_
        uoff.append(at)
        at += len(w) + int(rng.integers(0, 7))            # a few guard bytes between the outputs
    out = gpu_inflate(bytes(comp), blocks, uoff=uoff).tobytes()
    ref = bytearray(len(out))
    for w, u in zip(want, uoff):
        ref[u:u + len(w)] = w
    if out != bytes(ref):
        bad = next(i for i, (w, u) in enumerate(zip(want, uoff)) if out[u:u + len(w)] != w) if any(out[u:u + len(w)] != w for w, u in zip(want, uoff)) else -1
        open(os.path.join(out_dir, f"inflate_{it}.bin"), "wb").write(streams[max(bad, 0)])
        raise Mismatch(f"inflate differs (stream {bad}; -1 = a guard byte was written)")
    # damaged streams, one at a time beside a good one: zlib's verdict is the expectation when it accepts; what zlib
    # refuses must be refused or at least stay inside its own output (the caller's CRC32 check sees the rest)
    n_bad = lenient = 0
    for _ in range(int(rng.integers(5, 40))):
        k = int(rng.integers(0, len(streams)))
        s = bytearray(streams[k])
        if not s:
            continue
        x = int(rng.integers(0, 4))
        if x == 0:
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        elif x == 1:
            s = s[:int(rng.integers(0, len(s)))]
        elif x == 2:
            p = int(rng.integers(0, len(s)))
            s[p:p + 8] = rng.integers(0, 256, size=min(8, len(s) - p), dtype=np.uint8).tobytes()
        else:
            s[0] ^= int(rng.integers(1, 8))               # block type / final bit
        isz = len(want[k]) if rng.random() < 0.8 else max(0, len(want[k]) + int(rng.integers(-3, 4)))
        ok, zout = _zlib_inflate(bytes(s), isz)
        good = streams[0]
        cbuf = good + bytes(s)
        try:
            o = gpu_inflate(cbuf, [(0, len(good), len(want[0])), (len(good), len(s), isz)], uoff=[8, 8 + len(want[0]) + 8]).tobytes()
            gpu_ok = True
        except SvdssError as e:
            gpu_ok = False
            if e.bad_block != 1:
                raise Mismatch(f"a damaged stream was reported as block {e.bad_block}")
        if ok:
            if not gpu_ok or o[8 + len(want[0]) + 8:8 + len(want[0]) + 8 + isz] != zout[:isz]:
                open(os.path.join(out_dir, f"inflate_damaged_{it}.bin"), "wb").write(bytes(s))
                raise Mismatch(f"zlib accepts a damaged stream ({isz} bytes) and the GPU {'differs' if gpu_ok else 'refuses'}")
        elif gpu_ok:
            lenient += 1
            if o[:8] != b"\0" * 8 or o[8:8 + len(want[0])] != want[0] or o[8 + len(want[0]):8 + len(want[0]) + 8] != b"\0" * 8:
                raise Mismatch("a damaged stream wrote outside its own output")
        n_bad += 1
    return sum(len(w) for w in want), f"{len(streams)} streams, {n_bad} damaged ({lenient} accepted by the GPU only)"


def fuzz_deflate(rng, out_dir, it):
    from svdss_amd.bgzf import gpu_deflate
    from tests.test_deflate_gpu import members
    bb = int(rng.choice([0xff00, 0xff00, 0x8000, 4096, 1000, 64, 0xff00 - 1]))
    data = _mixture(rng, int(rng.integers(1, 40 * bb if bb >= 4096 else 200 * bb)))
    dense = rng.random() < 0.5
    stream = gpu_deflate(data, bb, dense=bool(dense))
    ms = members(stream)
    if len(ms) != (len(data) + bb - 1) // bb:
        raise Mismatch("member count")
    for i, (raw, crc, isize) in enumerate(ms):
        d = zlib.decompressobj(-15)
        got = d.decompress(raw) + d.flush()
        w = data[i * bb:(i + 1) * bb]
        if not d.eof or d.unused_data or got != w or isize != len(w) or crc != (zlib.crc32(w) & 0xffffffff) or 18 + len(raw) + 8 > 65536:
            open(os.path.join(out_dir, f"deflate_{it}.bin"), "wb").write(w)
            raise Mismatch(f"deflate member {i} of {len(ms)} (block_bytes {bb}, dense {dense}) does not inflate to its input")
    return len(data), f"{len(ms)} members, ratio {len(stream) / max(1, len(data)):.3f}"


# ------------------------------------------------------------------------------------------------ placement / index

def fuzz_place(rng, out_dir, it):
    from tests.mirror.clusterer import Clusterer
    from tests.test_place_gpu import _place, _random_case
    chroms = []
    for n in (int(rng.integers(2000, 40000)), int(rng.integers(800, 9000))):
        c = rng.integers(1, 5, size=n).astype(np.uint8)
        for _ in range(int(rng.integers(0, 8))):
            a = int(rng.integers(0, max(1, n - 700)))
            c[a:a + 600] = np.tile(rng.integers(1, 5, size=int(rng.integers(1, 5))).astype(np.uint8), 600)[:len(c[a:a + 600])]
        if rng.random() < 0.3:
            a = int(rng.integers(0, n - 50)); c[a:a + 40] = 5
        chroms.append(synth.to_ascii(c))
    names = ["c0", "c1"]
    alns, lists = [], []
    for k in range(int(rng.integers(50, 800))):
        a, sfs = _random_case(rng, chroms, k)
        alns.append(a); lists.append(sfs)
    got, stats = _place(chroms, alns, lists)
    cl = Clusterer({a.qname: s for a, s in zip(alns, lists)}, dict(zip(names, chroms)), names, threads=1)
    for a, sfs, g in zip(alns, lists, got):
        want = cl.extend_alignment(a)
        if [(x.rs, x.re, x.qs, x.qe) for x in want] != [t[:4] for t in g] or [x.htag for x in want] != [sfs[t[4]][2] for t in g]:
            raise Mismatch(f"placement differs: {a.qname} pos {a.pos} cigar {a.cigar} sfs {sfs}")
    if stats != [cl.unplaced, cl.s_unplaced, cl.e_unplaced, cl.unknown]:
        raise Mismatch(f"placement counters {stats} vs {[cl.unplaced, cl.s_unplaced, cl.e_unplaced, cl.unknown]}")
    return sum(len(s) for s in lists), len(alns)


def fuzz_index(rng, out_dir, it):
    """the index built in HBM against the host builder's (byte-equal files), the rld0 file written and read back, and
    the index imported from the rld0 file alone against the oracle's search"""
    import tempfile
    contigs = [_weird_contig(rng, int(rng.choice([200, 5000, 60000, 250000]))) for _ in range(int(rng.integers(1, 6)))]
    env = {}
    if rng.random() < 0.6:
        env["SVDSS_SA_PIECE"] = str(int(rng.choice([16, 300, 5000, 100000])))
    if rng.random() < 0.3:
        env["SVDSS_FORCE_SA64"] = "1"
    for k_ in ("SVDSS_SA_PIECE", "SVDSS_FORCE_SA64", "SVDSS_INDEX_CPU", "SVDSS_KMER"):
        os.environ.pop(k_, None)
    with tempfile.TemporaryDirectory() as td:
        try:
            os.environ.update(env)
            os.environ["SVDSS_INDEX_CPU"] = "1"
            svdss_amd.FMDIndex.build(contigs).save(os.path.join(td, "cpu.idx"))
            del os.environ["SVDSS_INDEX_CPU"]
            g = svdss_amd.FMDIndex.build(contigs, device=0)
            g.save(os.path.join(td, "gpu.idx"))
            if open(os.path.join(td, "gpu.idx"), "rb").read() != open(os.path.join(td, "cpu.idx"), "rb").read():
                raise Mismatch(f"the HBM-built index differs from the host builder's (env {env}): " + _dump(out_dir, f"index_{it}", contigs=contigs, env=str(env)))
            g.save_fmd(os.path.join(td, "x.fmd"))
            g.close()
            for k_ in env:
                os.environ.pop(k_, None)
            imp = svdss_amd.FMDIndex.load(os.path.join(td, "x.fmd")).to_device(0)       # no sidecar: decode, recover, rebuild
            fm = O.OracleFMD.build(contigs)
            # (the import keeps one strand of every pair in the order of the file's sentinels: another valid BWT of the
            # same collection, so it is checked against its own text and through the search, not byte for byte)
            v = imp.verify(1)
            if imp.size != fm.n or v["bad_order"] or v["bad_bwt"] or v["bad_range"] or v["bad_block"] or v["bad_dollar"]:
                raise Mismatch(f"the imported index fails its verification {v}: " + _dump(out_dir, f"index_{it}", contigs=contigs, env=str(env)))
            reads = _weird_reads(rng, contigs, 60)
            flat, offs = svdss_amd.pack_reads(reads)
            pp = svdss_amd.PingPong(imp, assemble=True)
            got = pp.ping_pong_search(flat, offs)
            pp.close()
            c, q, l, e = fm.search_batch(flat, offs, True)
            if not ((got.counts == c).all() and (got.n_ext == e).all() and (got.qs == q).all() and (got.len == l).all()):
                raise Mismatch("search on the imported index differs: " + _dump(out_dir, f"index_{it}", contigs=contigs, reads=reads))
            imp.close()
        finally:
            for k_ in list(env) + ["SVDSS_INDEX_CPU"]:
                os.environ.pop(k_, None)
    return sum(len(c) for c in contigs), len(contigs)


FUZZERS = {"search": fuzz_search, "poa": fuzz_poa, "extd2": fuzz_extd2, "ratio": fuzz_ratio, "inflate": fuzz_inflate,
           "deflate": fuzz_deflate, "place": fuzz_place, "index": fuzz_index}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default=",".join(FUZZERS))
    ap.add_argument("--minutes", type=float, default=2.0, help="per fuzzer")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/fuzz")
    ap.add_argument("--max-iterations", type=int, default=0)
    ap.add_argument("--replay", default="", help="a poa_*.npz written by an earlier run: that cluster alone, again")
    args = ap.parse_args()
    if args.replay:
        from svdss_amd import calldp
        d = np.load(args.replay)
        flat, off = d["reads_flat"], d["reads_off"]
        reads = [np.ascontiguousarray(flat[off[i]:off[i + 1]], dtype=np.uint8) for i in range(len(off) - 1)]
        want = bytes(LET[O.poa_consensus(reads)]).decode()
        got, stats = calldp.run_poa([reads])
        print(f"[replay] {len(reads)} reads of {[len(r) for r in reads]}: consensus {len(got[0])} vs the oracle's {len(want)}: "
              f"{'same' if got[0] == want else 'DIFFERENT'} ({stats})")
        # which reads does it take?  every prefix of the cluster, and the cluster without its empty reads
        for n in range(1, len(reads) + 1):
            g, _ = calldp.run_poa([reads[:n]])
            w = bytes(LET[O.poa_consensus(reads[:n])]).decode()
            print(f"[replay]   first {n} read(s): {'same' if g[0] == w else 'DIFFERENT'}")
        ne = [r for r in reads if len(r)]
        g, _ = calldp.run_poa([ne])
        print(f"[replay]   without empty reads: {'same' if g[0] == bytes(LET[O.poa_consensus(ne)]).decode() else 'DIFFERENT'}")
        sys.exit(0 if got[0] == want else 1)
    os.makedirs(args.out, exist_ok=True)
    rc = 0
    for name in args.what.split(","):
        fn = FUZZERS[name]
        t0 = time.time()
        it = units = 0
        last = ""
        try:
            while time.time() - t0 < args.minutes * 60 and (not args.max_iterations or it < args.max_iterations):
                rng = np.random.default_rng([args.seed, it, sum(map(ord, name))])
                u, last = fn(rng, args.out, it)
                units += u
                it += 1
            print(f"[fuzz] {name}: {it} iterations, {units} units, {time.time() - t0:.0f} s, seed {args.seed}: no difference (last: {last})", flush=True)
        except Mismatch as e:
            print(f"[fuzz] {name}: MISMATCH at iteration {it} (seed {args.seed}): {e}", flush=True)
            rc = 1
        except Exception as e:                            # an error of the library is a finding too
            import traceback
            traceback.print_exc()
            print(f"[fuzz] {name}: ERROR at iteration {it} (seed {args.seed}): {type(e).__name__}: {e}", flush=True)
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
