"""The segmented search kernel under concurrent launches.

Root cause of round 4's "segmented kernel with the BS code compiled in gives run-to-run different text": emit() stores the
first records of a segment with a 16-byte `global_store_dwordx4 ... sc1` in inline asm; a vector-memory store of more than
8 bytes reads its data registers after it has issued, the compiler's hazard recogniser cannot see inside the asm, and in
that instantiation the register allocation put `v_mov_b32 v3, 1` one instruction behind the store of v[2:5]: the record's
length reached memory as 1 or as itself depending on the load of the memory pipeline.  The wait states are now part of the
asm (csrc/sfs_search.hip).  `make -C svdss_amd/csrc hazard` + tools/seg_stress_bam.py still show the old behaviour
(profiles/r05c_segmented_store_hazard.txt: 8 distinct outputs in 8 runs; this tree: 1).

Here: round 4's scenario through the binary (a BAM of many 1-MB device batches, six feeding threads, eight segments per
read) on both instantiations, and six threads x 40 launches in one process against the one-lane-per-read result."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import svdss_amd
from tests import test_bam_device_gpu as T
from tests.common import BIN, small_workload
from tools import seg_stress

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bs", ["0", "1"])
def test_concurrent_segmented_launches_in_one_process(monkeypatch, bs):
    monkeypatch.setenv("SVDSS_BS", bs)
    bad, n, first = seg_stress.run(threads=6, repeats=40, segments=8)
    assert n == 240 and bad == 0, first


def test_search_bam_with_six_feeders_and_eight_segments_is_one_text(tmp_path):
    ref, hap, svs, flat, offs = small_workload(seed=91, n_reads=400, read_len=1500, ref_lens=(150000,))
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(400)]
    reads[14] = np.concatenate([reads[14], reads[15], reads[16], reads[17]])
    names = [f"m64/{(i * 37) % 400}/ccs" for i in range(400)]
    rng = np.random.default_rng(23)
    recs = []
    for rep in range(8):
        r, _ = T._records([f"{n}/{rep}" for n in names], reads, rng, decoys=(rep % 2 == 0))
        recs += r
    bam, fmd = str(tmp_path / "reads.bam"), str(tmp_path / "ref.fmd")
    open(bam, "wb").write(T._bgzf_levels(T._raw_bam([("chr1", 150000)], recs), rng, block=60000))
    ix.save(fmd)
    del ix
    base = {"SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SLAB_KB": "64"}
    seen = set()
    for env, runs in [(dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="1"), 1), (dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="8"), 12),
                      (dict(base, SVDSS_BS="0", SVDSS_SEGMENTS="8"), 12)]:
        for _ in range(runs):
            r = subprocess.run([BIN, "search", "--index", fmd, "--bam", bam, "--threads", "4", "--bsize", "100"], capture_output=True,
                               text=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-400:]
            seen.add(hashlib.md5(r.stdout.encode()).hexdigest())
    assert len(seen) == 1, seen
