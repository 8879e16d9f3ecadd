#!/usr/bin/env python3
"""GPU index builder (csrc/index_gpu.hip) against the host builder, plus timings at chr20 / GRCh38 lengths.

  python tools/index_gpu_check.py small          # byte-identical files, many piece sizes
  python tools/index_gpu_check.py time 64444167  # one contig of that length: build time, resident
  python tools/index_gpu_check.py wg             # 24 contigs with GRCh38 primary lengths
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdss_amd  # noqa: E402
from svdss_amd import synth  # noqa: E402


def small():
    refs = {
        "repeats+N": synth.make_reference([180000, 90000, 700], seed=33, repeat_frac=0.5, divergence=0.0005,
                                          n_runs=(500, 30)) + [np.tile(np.array([1, 2, 3, 4], np.uint8), 3000)],
        "tiny": [np.array([1, 2, 3], np.uint8), np.array([4], np.uint8), np.array([1, 1, 1, 1, 1, 1, 1, 1, 1], np.uint8)],
        "polyA": [np.full(5000, 1, np.uint8), synth.make_reference([3000], seed=2)[0]],
    }
    d = tempfile.mkdtemp()
    for name, ref in refs.items():
        os.environ["SVDSS_INDEX_CPU"] = "1"
        cpu = svdss_amd.FMDIndex.build(ref)
        cpu.save(f"{d}/cpu.fmd")
        want = open(f"{d}/cpu.fmd", "rb").read()
        del os.environ["SVDSS_INDEX_CPU"]
        for piece in (None, "100000", "20000", "3000"):
            for wide in (False, True):
                if piece:
                    os.environ["SVDSS_SA_PIECE"] = piece
                else:
                    os.environ.pop("SVDSS_SA_PIECE", None)
                if wide:
                    os.environ["SVDSS_FORCE_SA64"] = "1"
                else:
                    os.environ.pop("SVDSS_FORCE_SA64", None)
                if wide:
                    os.environ["SVDSS_INDEX_CPU"] = "1"
                    w = svdss_amd.FMDIndex.build(ref)
                    w.save(f"{d}/cpu64.fmd")
                    del os.environ["SVDSS_INDEX_CPU"]
                    expect = open(f"{d}/cpu64.fmd", "rb").read()
                else:
                    expect = want
                os.environ["SVDSS_INDEX_VERBOSE"] = "1"
                g = svdss_amd.FMDIndex.build(ref, device=0)
                g.save(f"{d}/gpu.fmd")
                got = open(f"{d}/gpu.fmd", "rb").read()
                ok = got == expect
                print(f"{name:10s} piece={piece} wide={wide}: {'identical' if ok else 'DIFFERENT'} "
                      f"({len(got)} bytes, K={g.kmer_k})", flush=True)
                if not ok:
                    a, b = np.frombuffer(got, np.uint8), np.frombuffer(expect, np.uint8)
                    n = g.size
                    nb = n // 128 + 1
                    secs = [("header", 0, 104), ("blocks", 104, 104 + 64 * nb)]
                    o = 104 + 64 * nb
                    secs.append(("dollar", o, o + 16 * len(ref)))
                    o += 16 * len(ref)
                    secs.append(("text", o, o + n))
                    secs.append(("sa", o + n, len(got)))
                    for nm, lo, hi in secs:
                        d_ = np.nonzero(a[lo:hi] != b[lo:hi])[0] if len(a) == len(b) else []
                        print(f"   {nm}: {len(d_)} differing bytes" + (f", first at +{d_[0]}" if len(d_) else ""))
                    raise SystemExit(1)
    os.environ.pop("SVDSS_SA_PIECE", None)
    os.environ.pop("SVDSS_FORCE_SA64", None)
    print("small: ok")


def timed(lens):
    t0 = time.time()
    ref = synth.make_reference(lens, seed=11)
    t1 = time.time()
    print(f"reference: {sum(lens)} bp in {len(lens)} contigs, generated in {t1 - t0:.1f} s", flush=True)
    os.environ["SVDSS_INDEX_VERBOSE"] = "1"
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    t2 = time.time()
    print(f"svdss_index_build_device: {t2 - t1:.1f} s, n = {ix.size}, K = {ix.kmer_k}, "
          f"{ix.device_bytes / 1e9:.1f} GB resident", flush=True)
    # spot check against direct counting: patterns cut from the reference must occur, interval sizes via the host blocks
    rng = np.random.default_rng(1)
    for _ in range(200):
        ci = int(rng.integers(0, len(ref)))
        s = int(rng.integers(0, len(ref[ci]) - 40))
        w = ref[ci][s:s + 32]
        assert ix.count(w) >= 1 and ix.count(synth.revcomp(w)) == ix.count(w)
    assert (np.diff(ix.acc) >= 0).all() and ix.acc[-1] == ix.size
    print("spot checks ok", flush=True)
    return ix


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    if mode == "small":
        small()
    elif mode == "time":
        timed([int(sys.argv[2])])
    elif mode == "wg":
        from bench import GRCH38_PRIMARY
        timed(GRCH38_PRIMARY)
