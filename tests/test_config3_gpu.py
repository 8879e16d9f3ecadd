"""BASELINE config 3: chr20-length reference, 10x HiFi-length reads (42,963 x 15 kb), SFS search + clustering + POA +
realignment end to end through the `SVDSS` binaries, VCF checked against the implanted truth (heterozygous and
homozygous SVs) and -- on the reads around a few of the SVs, where the Python mirror of the reference's host logic
finishes in seconds -- byte for byte against svdss_amd.caller.call."""
import os
import subprocess

import pytest

from svdss_amd import synth
from tests.mirror import bamio, caller
from tests.common import ROOT
from tools import e2e_call

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


def _chain(work, fa, bam, threads="4"):
    fmd = os.path.join(work, "ref.fa.fmd")
    if not os.path.exists(fmd):
        r = subprocess.run([BIN, "index", "-t", "16", "-d", fa, "-o", fmd], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([BIN, "search", "--index", fmd, "--bam", bam, "--threads", threads], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sfs = bam + ".sfs"
    with open(sfs, "w") as fh:
        fh.write(r.stdout)
    r2 = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", sfs, "--threads", threads,
                         "--min-sv-length", "50"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    return r.stdout, r2.stdout


def test_chr20_10x_end_to_end(tmp_path):
    work = str(tmp_path)
    L, ref_bp = 15000, 64_444_167
    fa, bam, svs, het, recs, hdr, ref = e2e_call.write_dataset(work, ref_bp, 64, 10.0, L, seed=11, het_every=2, max_len=2000)
    assert 42_000 < len(recs) < 43_500 and sum(het) == 32
    sfs_text, vcf = _chain(work, fa, bam)
    called = e2e_call.parse_vcf(vcf)
    truth = [(s.pos, s.kind, s.length) for s in svs]
    # every implanted SV is called once, at its position (insertion points may shift within a repeat of the flanks),
    # with its type and length; nothing else is called
    for (p, k, l), h in zip(truth, het):
        hits = [c for c in called if c[1] == k and c[2] == l and abs(c[0] - p) <= 12]
        assert len(hits) == 1, (p, k, l, h, [(c[0], c[1], c[2]) for c in called if abs(c[0] - p) < 3000])
    assert len(called) == len(truth)
    # (placement on the GPU, the default, and on the host give the same VCF at this size too)
    r_host = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4",
                             "--min-sv-length", "50"], capture_output=True, text=True, env=dict(os.environ, SVDSS_PLACE_HOST="1"))
    assert r_host.returncode == 0 and r_host.stdout == vcf
    # (the BAM read through the device path -- the default: svdss_bam_select_run keeps the records of reads with SFS in pass 1
    # and, without an index, those that overlap a cluster in pass 2 -- and through the host reader with its record cache)
    r_dev = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50",
                            "--verbose"], capture_output=True, text=True, env=dict(os.environ, SVDSS_BAM_BATCH_MB="8", SVDSS_BAM_SLAB_KB="512"))
    assert r_dev.returncode == 0 and r_dev.stdout == vcf, r_dev.stderr[-500:]
    # round 6: ONE pass over the file -- what pass 2 needs of every record stays in HBM while pass 1 runs (svdss_bam_store_t);
    # in arenas of 1 MB (dozens); with a store too small for the file (pass 2 reads the file again); switched off
    assert "pass 2 from the records kept in HBM" in r_dev.stderr
    for env, what in (({"SVDSS_STORE_ARENA_MB": "1", "SVDSS_BAM_BATCH_MB": "2", "SVDSS_BAM_SLAB_KB": "256"}, "pass 2 from the records kept in HBM"),
                      ({"SVDSS_CALL_STORE_MB": "3", "SVDSS_STORE_ARENA_MB": "1", "SVDSS_BAM_BATCH_MB": "2", "SVDSS_BAM_SLAB_KB": "256"}, "the record store is incomplete"),
                      ({"SVDSS_CALL_STORE": "0"}, None)):
        r_st = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50", "--verbose"],
                              capture_output=True, text=True, env=dict(os.environ, **env))
        assert r_st.returncode == 0 and r_st.stdout == vcf, (env, r_st.stderr[-500:])
        assert (what in r_st.stderr) if what else ("record store is incomplete" not in r_st.stderr and "kept in HBM" not in r_st.stderr), (env, r_st.stderr[-800:])
    r_hr = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50"],
                          capture_output=True, text=True, env=dict(os.environ, SVDSS_BAM_DEVICE="0"))
    assert r_hr.returncode == 0 and r_hr.stdout == vcf
    # --gpus 3 (three filters / feeding groups / DP shards on the box's one GPU): the batches of the BAM go to whichever
    # shard is free, the VCF is the same
    r_g3 = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50", "--gpus", "3"],
                          capture_output=True, text=True, env=dict(os.environ, SVDSS_GPUS_OVERSUBSCRIBE="1", SVDSS_BAM_BATCH_MB="8", SVDSS_BAM_SLAB_KB="512"))
    assert r_g3.returncode == 0 and r_g3.stdout == vcf, r_g3.stderr[-400:]
    # round 6: --gpus N cuts the file into N regions, each with its own scanner / batcher / feeders / record stream / filter /
    # record store (ShardedBamSelect); a region's first record is guessed and proved at the seam.  SVDSS_REGION_TEST: 1 =
    # every guess is no record (the regions fail and run again from the region before), 2 = the seams do not fit (the
    # regions run again); with the stores (one pass) and without (two passes, both sharded)
    import re
    size_kb = os.path.getsize(bam) >> 10
    for n, knob, store in ((3, None, True), (4, "1", True), (2, "2", True), (7, None, True), (3, None, False), (4, "1", False)):
        env = {"SVDSS_GPUS_OVERSUBSCRIBE": "1", "SVDSS_BAM_BATCH_MB": "2", "SVDSS_BAM_SLAB_KB": "256", "SVDSS_REGION_MIN_KB": str(max(64, size_kb // 8)),
               "SVDSS_STORE_ARENA_MB": "4", "SVDSS_CALL_STORE_INITIAL_MB": "4"}
        if knob:
            env["SVDSS_REGION_TEST"] = knob
        if not store:
            env["SVDSS_CALL_STORE"] = "0"
            env["SVDSS_CALL_PASS2"] = "device"
        r_sh = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50", "--gpus", str(n),
                               "--verbose"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r_sh.returncode == 0 and r_sh.stdout == vcf, (n, knob, store, r_sh.stderr[-600:])
        m = re.search(r"(\d+) regions of the file, one per GPU: (\d+) seam\(s\) run, (\d+) region\(s\) run again", r_sh.stderr)
        assert m and int(m.group(1)) == n, (n, knob, r_sh.stderr[-800:])
        if knob is None:
            assert m.group(3) == "0"
        elif knob == "1":
            assert m.group(3) == str(n - 1)
        else:
            assert int(m.group(3)) >= 1
        assert ("pass 2 from the records kept in HBM" in r_sh.stderr) == store
    # the second BAM pass without the records of pass 1 in memory: the whole file again, or -- with a BAI index beside
    # the file, as the reference requires -- only the chunks the index names for the cluster regions (records here
    # straddle BGZF blocks, chunks start and end inside blocks): the same VCF
    for with_bai in (False, "csi", True):
        if with_bai:
            from tests import bam_writer
            with open(bam, "rb") as fh:
                idx = bam_writer.csi(fh.read(), 14, 6) if with_bai == "csi" else bam_writer.bai(fh.read())
            if with_bai is True:
                os.remove(bam + ".csi")
            with open(bam + (".csi" if with_bai == "csi" else ".bai"), "wb") as fh:
                fh.write(idx)
        r_p2 = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4",
                               "--min-sv-length", "50", "--verbose"], capture_output=True, text=True,
                              env=dict(os.environ, SVDSS_CALL_CACHE_GB="0", SVDSS_CALL_PASS2="bai", SVDSS_CALL_STORE="0"))
        assert r_p2.returncode == 0, r_p2.stderr[-500:]
        assert ("through the BAI index" in r_p2.stderr) == (with_bai is True)
        assert ("through the CSI index" in r_p2.stderr) == (with_bai == "csi")
        assert r_p2.stdout == vcf
        if with_bai:
            # the chunks the index names scanned by one host thread and by seven (runs of consecutive chunks taken in turn,
            # applied in file order): the same VCF
            for thr in ("1", "7"):
                r_t = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50"],
                                     capture_output=True, text=True,
                                     env=dict(os.environ, SVDSS_CALL_CACHE_GB="0", SVDSS_CALL_PASS2="bai", SVDSS_CALL_PASS2_THREADS=thr, SVDSS_CALL_STORE="0"))
                assert r_t.returncode == 0 and r_t.stdout == vcf, (thr, r_t.stderr[-300:])
        # (and the host reader without its cache, with and without the index; with the index present the device path may
        # still read the whole file when the chunks the index names are a large part of it: SVDSS_CALL_PASS2=device)
        for env in ({"SVDSS_BAM_DEVICE": "0"}, {"SVDSS_CALL_PASS2": "device"}):
            r_p3 = subprocess.run([BIN, "call", "--reference", fa, "--bam", bam, "--sfs", bam + ".sfs", "--threads", "4", "--min-sv-length", "50"],
                                  capture_output=True, text=True, env=dict(os.environ, SVDSS_CALL_CACHE_GB="0", SVDSS_CALL_STORE="0", **env))
            assert r_p3.returncode == 0 and r_p3.stdout == vcf, (env, r_p3.stderr[-300:])
    os.remove(bam + ".bai")
    # the reads around four SVs (two heterozygous, two homozygous): the same chain on that sub-BAM, and the Python mirror
    # of the host logic (clusterer.cpp / caller.cpp restated in svdss_amd/) on the same inputs -> the same VCF bytes
    pick = [k for k in range(len(svs)) if het[k]][:2] + [k for k in range(len(svs)) if not het[k]][:2]
    win = [(svs[k].pos - 25_000, svs[k].pos + 25_000) for k in pick]
    sub = [r[2] for r in recs if any(r[0] < b and r[1] > a for a, b in win)]
    assert 100 < len(sub) < 600
    sub_bam = os.path.join(work, "sub.bam")
    e2e_call.write_bam_records(sub_bam, hdr, sub)
    sub_sfs, sub_vcf = _chain(work, fa, sub_bam)
    ref_names, ref_lens, alns = bamio.read_bam(sub_bam)
    chromosomes = {"chrS": synth.to_ascii(ref[0])}
    mirror_vcf, info = caller.call(alns, sub_sfs, chromosomes, list(zip(ref_names, ref_lens)), ref_names, threads=4,
                                   min_sv_length=50)
    assert sub_vcf == mirror_vcf
    sub_called = e2e_call.parse_vcf(sub_vcf)
    assert len(sub_called) == 4
    full_rows = {(c[0], c[1], c[2]): c[3] for c in called}
    for c in sub_called:       # and those rows are the rows of the full run (same reads in the cluster, same consensus)
        f = full_rows[(c[0], c[1], c[2])]
        assert f[:5] == c[3][:5]
