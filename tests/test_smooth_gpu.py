"""K6: `SVDSS smooth` with the CIGAR walk on the GPU (svdss_smooth_batch, csrc/place.hip) -- the default
when a GPU is present -- against the Python mirror of smoother.cpp and against the host code of the same binary
(SVDSS_SMOOTH_HOST=1): the same BAM bytes."""
import os
import subprocess

import numpy as np
import pytest

from svdss_amd import synth
from tests import bam_writer, test_smooth
from tests.common import ROOT
from tests.pipeline_sim import add_errors, simulate

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


def test_gpu_smooth_matches_the_mirror(tmp_path):
    test_smooth.test_smooth_cli_matches_mirror(tmp_path)            # the binary takes the GPU path here


def test_gpu_smooth_handles_inconsistent_records(tmp_path):
    test_smooth.test_inconsistent_records_pass_through_with_xf3(tmp_path)


def test_gpu_and_host_smooth_write_the_same_bam(tmp_path):
    ref, svs, reads = simulate(ref_lens=(300000, 90000), n_svs=12, coverage=12, read_len=9000, seed=18)
    rng = np.random.default_rng(3)
    names = ["c0", "c1"]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for n, c in zip(names, ref):
            fh.write(f">{n}\n{synth.to_ascii(c)}\n")
    recs = []
    for k, (n, tid, pos, cig, seq, hp) in enumerate(reads):
        s2, c2 = add_errors(seq, cig, rng, 0.05 if k % 19 == 0 else 0.006)
        qual = bytes(rng.integers(1, 60, size=len(s2)).astype(np.uint8))
        recs.append(bam_writer.record(n, 16 if k % 2 else 0, tid, pos, 60, c2, s2, [("HP", "C", hp)] if hp else [], qual))
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    outs = {}
    for tag, env in (("gpu", {}), ("host", {"SVDSS_SMOOTH_HOST": "1"})):
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4"], capture_output=True,
                           timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = r.stdout
    assert len(outs["gpu"]) > 100000 and outs["gpu"] == outs["host"]


def test_device_path_writes_the_bytes_of_the_host_paths(tmp_path):
    """Round 5: records filtered, measured, smoothed, rebuilt and deflated in HBM (csrc/bam_smooth.inc) against the host
    pipeline with the GPU walk (SVDSS_BAM_DEVICE=0) and the host code (SVDSS_SMOOTH_HOST=1): the same bytes -- with XF tags
    already present in every integer type and as a string, records the filters drop, CIGARs that do not fit, reads on a
    contig the FASTA does not have, and device batches of one megabyte handed round six feeding threads (the output stream's
    turn: BGZF blocks that begin in one batch and end in the next)."""
    ref, svs, reads = simulate(ref_lens=(300000, 90000, 20000), n_svs=12, coverage=10, read_len=7000, seed=28)
    rng = np.random.default_rng(5)
    names = ["c0", "c1", "c2"]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for n, c in zip(names[:2], ref[:2]):          # (c2 is not in the FASTA: its records are dropped)
            fh.write(f">{n}\n{synth.to_ascii(c)}\n")
    recs = []
    for k, (n, tid, pos, cig, seq, hp) in enumerate(reads):
        s2, c2 = add_errors(seq, cig, rng, 0.05 if k % 17 == 0 else 0.006)
        qual = bytes(rng.integers(1, 60, size=len(s2)).astype(np.uint8))
        tags = [("HP", "C", hp)] if hp else []
        if k % 5 == 1:
            tags.append(("XF", "C", 9))
        elif k % 5 == 2:
            tags = [("XF", "i", 70000)] + tags + [("ZZ", "Z", "after")]
        elif k % 5 == 3:
            tags.append(("XF", "Z", "text"))
        flag = 16 if k % 2 else 0
        if k % 23 == 0:
            flag |= 256
        if k % 29 == 0:
            flag |= 2048
        mapq = 5 if k % 31 == 0 else 60
        if k % 37 == 0:
            c2 = c2[:-1] + [(c2[-1][0], c2[-1][1] + 3)]           # a CIGAR that does not add up: XF = 3
        recs.append(bam_writer.record(n, flag, tid, pos, mapq, c2, s2, tags, qual))
    bam = tmp_path / "in.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    outs = {}
    for tag, env in (("device", {}), ("device small batches", {"SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SLAB_KB": "64"}),
                     ("host + gpu walk", {"SVDSS_BAM_DEVICE": "0"}), ("host", {"SVDSS_SMOOTH_HOST": "1"})):
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4", "--min-mapq", "20"], capture_output=True,
                           timeout=900, env=dict(os.environ, SVDSS_DEBUG="1", **env))
        assert r.returncode == 0, r.stderr.decode()
        outs[tag] = r.stdout
        if tag == "device":
            assert b"device path" in r.stderr
    assert len(outs["host"]) > 100000
    for tag in outs:
        assert outs[tag] == outs["host"], tag
    # --gpus N (round 6): the file's regions, one per GPU (two or three of them on the one GPU of this box).  The records and
    # their order are the single GPU's; the BGZF members are not cut at the same bytes (a region ends with a short member, a
    # seam's record is a member of its own): the inflated streams are compared.  SVDSS_REGION_TEST makes the guesses at the
    # regions' starts fail (1: no record there; 2: a head one byte short): those regions run again from the known carry.
    import gzip
    import re
    want = gzip.decompress(outs["host"])
    small = {"SVDSS_GPUS_OVERSUBSCRIBE": "1", "SVDSS_REGION_MIN_KB": "256", "SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SLAB_KB": "64", "SVDSS_DEBUG": "1"}
    for gpus, env, reruns in (("2", {}, 0), ("3", {"SVDSS_SEARCH_FEEDERS": "2"}, 0), ("4", {"SVDSS_REGION_TEST": "1"}, 3),
                              ("2", {"SVDSS_REGION_TEST": "2"}, 1)):
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4", "--min-mapq", "20", "--gpus", gpus],
                           capture_output=True, timeout=900, env=dict(os.environ, **small, **env))
        assert r.returncode == 0, r.stderr.decode()
        m = re.search(rb"(\d+) regions on 1 GPU\(s\): (\d+) seam\(s\) proved, (\d+) region\(s\) run again", r.stderr)
        assert m and int(m.group(1)) == int(gpus) and int(m.group(3)) == reruns, r.stderr.decode()[-800:]
        assert r.stdout[-28:] == outs["host"][-28:] and gzip.decompress(r.stdout) == want, (gpus, env)
    # ... and into a regular file (the side-by-side writers)
    out = tmp_path / "sharded.bam"
    with open(out, "wb") as fh:
        r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam), "--threads", "4", "--min-mapq", "20", "--gpus", "3"],
                           stdout=fh, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, **small))
    assert r.returncode == 0, r.stderr.decode()
    assert gzip.decompress(out.read_bytes()) == want
