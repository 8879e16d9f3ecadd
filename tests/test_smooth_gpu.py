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
