"""End to end on synthetic data (SURVEY 8(d) config 1 shape, scaled down): FASTA -> `SVDSS index`
-> smoothed-style BAM -> `SVDSS search` (HIP) -> call (clusterer host logic + POA, realignment and
ratio kernels) -> VCF.  The VCF must recover the implanted SVs and nothing else."""
import os
import subprocess

import numpy as np
import pytest

from svdss_amd import synth
from tests.mirror import bamio, caller
from tests import bam_writer
from tests.common import ROOT
from tests.pipeline_sim import simulate

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "svdss_amd", "SVDSS")


@pytest.mark.parametrize("het", [0.0, 0.45])
def test_index_search_call_recovers_implanted_svs(tmp_path, het):
    ref, svs, reads = simulate(seed=5 if het == 0.0 else 6, het_fraction=het)
    names = ["chrA", "chrB"]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for n, c in zip(names, ref):
            fh.write(f">{n}\n{synth.to_ascii(c)}\n")
    fmd = tmp_path / "ref.fa.fmd"
    r = subprocess.run([BIN, "index", "-t", "8", "-d", str(fa), "-o", str(fmd)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    recs = [bam_writer.record(n, 0, tid, pos, 60, cig, seq, [("XF", "C", 0)]) for n, tid, pos, cig, seq, hp in reads]
    bam = tmp_path / "smoothed.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    r = subprocess.run([BIN, "search", "--index", str(fmd), "--bam", str(bam), "--threads", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sfs_text = r.stdout
    assert len(sfs_text) > 0
    ref_names, ref_lens, alns = bamio.read_bam(str(bam))
    assert ref_names == names and len(alns) == len(reads)
    chromosomes = {n: synth.to_ascii(c) for n, c in zip(names, ref)}
    vcf, info = caller.call(alns, sfs_text, chromosomes, list(zip(names, ref_lens)), ref_names, threads=4,
                            min_sv_length=50)
    body = [l.split("\t") for l in vcf.splitlines() if not l.startswith("#")]
    called = []
    for f in body:
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((f[0], int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"])), f))
    truth = [(names[s.contig], s.pos, s.kind, s.length) for s in svs]
    assert len(truth) == 8
    for chrom, pos, kind, length in truth:
        hits = [c for c in called if c[0] == chrom and c[2] == kind and c[3] == length and abs(c[1] - pos) <= 12]
        assert len(hits) == 1, (chrom, pos, kind, length, [(c[0], c[1], c[2], c[3]) for c in called])
        f = hits[0][4]
        # REF/ALT carry the anchor base; a deletion's REF is the deleted reference sequence
        if kind == "DEL":
            assert len(f[3]) == length + 1 and len(f[4]) == 1 and f[3] == chromosomes[chrom][hits[0][1] - 1:hits[0][1] + length]
        else:
            assert len(f[4]) == length + 1 and len(f[3]) == 1 and f[3] == chromosomes[chrom][hits[0][1] - 1]
    assert len(called) == len(truth)                      # no spurious calls
    assert info["subclusters"] >= len(truth)
    # the C++ CLI (`SVDSS call`, csrc/call_host.cpp) and the Python mirror give the same VCF bytes
    sfs_path = tmp_path / "specifics.txt"
    sfs_path.write_text(sfs_text)
    r = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs_path), "--threads", "4",
                        "--min-sv-length", "50", "--poa", str(tmp_path / "poa.sam"), "--clusters", str(tmp_path / "clusters.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == vcf
    # the second BAM pass three ways -- records of pass 1 kept in memory (above), the whole file read again, only the
    # file chunks a BAI index names for the cluster regions (sam_itr_querys in the reference): the same bytes
    clusters_text = (tmp_path / "clusters.txt").read_text()
    for with_bai in (False, True):
        if with_bai:
            (tmp_path / "smoothed.bam.bai").write_bytes(bam_writer.bai(bam.read_bytes()))
        r2 = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs_path), "--threads", "4",
                             "--min-sv-length", "50", "--clusters", str(tmp_path / "clusters2.txt"), "--verbose"],
                            capture_output=True, text=True, env=dict(os.environ, SVDSS_CALL_CACHE_GB="0", SVDSS_CALL_PASS2="bai", SVDSS_CALL_STORE="0"))
        assert r2.returncode == 0, r2.stderr
        # (SVDSS_CALL_PASS2=bai: with an index present the device path would otherwise weigh the chunks it names against
        # reading the whole file, tests/test_config3_gpu.py)
        assert ("through the BAI index" in r2.stderr) == with_bai, r2.stderr[-600:]
        assert r2.stdout == vcf and (tmp_path / "clusters2.txt").read_text() == clusters_text
    # side outputs (caller.cpp:65-75, clusterer.cpp:613-626)
    sam = (tmp_path / "poa.sam").read_text()
    assert sam == info["sam"]
    rows = [l.split("\t") for l in sam.splitlines() if not l.startswith("@")]
    assert len(rows) == info["subclusters"] and all(len(f) == 11 and f[1] == "0" and f[4] == "60" for f in rows)
    for f in rows:   # the CIGAR consumes the whole consensus
        import re
        assert sum(int(n) for n, op in re.findall(r"(\d+)([MID])", f[5]) if op in "MI") == len(f[9])
    cl_text = (tmp_path / "clusters.txt").read_text()
    assert cl_text == info["clusters_text"] and len(cl_text.splitlines()) == info["clusters"]
    # pass 2 from the records kept in memory (default) or from a second read of the BAM: the same bytes
    r2 = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs_path), "--threads", "4",
                         "--min-sv-length", "50"], capture_output=True, text=True, env=dict(os.environ, SVDSS_CALL_CACHE_GB="0"))
    assert r2.returncode == 0 and r2.stdout == vcf
    # --gpus N: POA and realignment batches shard by sub-cluster index, the rows come back in order, dedup and chain
    # filter run once on all of them: the same VCF and SAM bytes
    r3 = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs_path), "--threads", "4",
                         "--min-sv-length", "50", "--gpus", "3", "--poa", str(tmp_path / "poa3.sam")], capture_output=True,
                        text=True, env=dict(os.environ, SVDSS_GPUS_OVERSUBSCRIBE="1"))
    assert r3.returncode == 0, r3.stderr
    assert r3.stdout == vcf and (tmp_path / "poa3.sam").read_text() == sam
    # SFS placement runs on the GPU by default (csrc/place.hip); the host code of call_host.cpp gives the same bytes
    r4 = subprocess.run([BIN, "call", "--reference", str(fa), "--bam", str(bam), "--sfs", str(sfs_path), "--threads", "4",
                         "--min-sv-length", "50", "--clusters", str(tmp_path / "clusters_host.txt")], capture_output=True,
                        text=True, env=dict(os.environ, SVDSS_PLACE_HOST="1"))
    assert r4.returncode == 0, r4.stderr
    assert r4.stdout == vcf and (tmp_path / "clusters_host.txt").read_text() == cl_text


def test_run_svdss_chain_with_raw_reads(tmp_path):
    """The chain run_svdss executes (/root/reference/run_svdss:136-178): index -> smooth -> search -> call,
    all four through the `SVDSS` binary, on reads WITH sequencing errors (0.5 %, truthful CIGARs)."""
    from tests.pipeline_sim import add_errors
    ref, svs, reads = simulate(seed=9, coverage=30)
    rng = np.random.default_rng(4)
    names = ["chrA", "chrB"]
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for n, c in zip(names, ref):
            fh.write(f">{n}\n{synth.to_ascii(c)}\n")
    recs = []
    for n, tid, pos, cig, seq, hp in reads:
        s2, c2 = add_errors(seq, cig, rng, 0.005)
        recs.append(bam_writer.record(n, 0, tid, pos, 60, c2, s2))
    bam = tmp_path / "reads.bam"
    bam.write_bytes(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
    fmd = tmp_path / "ref.fa.fmd"
    assert subprocess.run([BIN, "index", "-t", "8", "-d", str(fa), "-o", str(fmd)], capture_output=True).returncode == 0
    smoothed = tmp_path / "smoothed.bam"
    with open(smoothed, "wb") as fh:
        r = subprocess.run([BIN, "smooth", "--threads", "4", "--min-mapq", "20", "--accp", "0.98", "--reference", str(fa),
                            "--bam", str(bam)], stdout=fh, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([BIN, "search", "--threads", "4", "--index", str(fmd), "--bam", str(smoothed)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sfs = tmp_path / "specifics.txt"
    sfs.write_text(r.stdout)
    # reads that carry no long indel are tagged XF=2 by smooth and skipped by search (putative filter)
    n_reads_with_sfs = len([l for l in r.stdout.splitlines() if not l.startswith("*")])
    assert 0 < n_reads_with_sfs < len(reads)
    r = subprocess.run([BIN, "call", "--threads", "4", "--min-cluster-weight", "2", "--min-sv-length", "50", "--min-mapq", "20",
                        "--reference", str(fa), "--bam", str(smoothed), "--sfs", str(sfs)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    called = []
    for line in r.stdout.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((f[0], int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"]))))
    truth = [(names[s.contig], s.pos, s.kind, s.length) for s in svs]
    for chrom, pos, kind, length in truth:
        hits = [c for c in called if c[0] == chrom and c[2] == kind and abs(c[3] - length) <= 2 and abs(c[1] - pos) <= 15]
        assert len(hits) == 1, (chrom, pos, kind, length, called)
    assert len(called) == len(truth)


def test_config1_shape_1k_reads_10mb_full_chain(tmp_path):
    """BASELINE config 1 at its own size: 1,000 reads of 15 kb with 0.5 % errors against a 10 Mb reference (1.5x),
    index -> smooth -> search -> call through the binaries (run_svdss:136-178).  At 1.5x only part of the implanted SVs
    is seen by two reads: what is called must be an implanted SV (type, length, position), and the VCF must equal the
    Python mirror of the reference's host logic (svdss_amd.caller.call: clusterer.cpp / caller.cpp restated) run on the
    same smoothed BAM and SFS file, byte for byte."""
    from tests.pipeline_sim import add_errors
    ref, svs, reads = simulate(ref_lens=(10_000_000,), n_svs=40, coverage=1.5, read_len=15000, seed=21)
    assert 990 <= len(reads) <= 1000
    rng = np.random.default_rng(22)
    fa = tmp_path / "ref.fa"
    fa.write_text(f">chr10M\n{synth.to_ascii(ref[0])}\n")
    recs = []
    for n, tid, pos, cig, seq, hp in reads:
        s2, c2 = add_errors(seq, cig, rng, 0.005)
        recs.append(bam_writer.record(n, 0, tid, pos, 60, c2, s2))
    bam = tmp_path / "reads.bam"
    bam.write_bytes(bam_writer.bam([("chr10M", len(ref[0]))], recs))
    fmd = tmp_path / "ref.fa.fmd"
    assert subprocess.run([BIN, "index", "-t", "8", "-d", str(fa), "-o", str(fmd)], capture_output=True).returncode == 0
    smoothed = tmp_path / "smoothed.bam"
    with open(smoothed, "wb") as fh:
        r = subprocess.run([BIN, "smooth", "--threads", "4", "--reference", str(fa), "--bam", str(bam)], stdout=fh,
                           stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([BIN, "search", "--threads", "4", "--index", str(fmd), "--bam", str(smoothed)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sfs = tmp_path / "specifics.txt"
    sfs.write_text(r.stdout)
    r = subprocess.run([BIN, "call", "--threads", "4", "--min-sv-length", "50", "--reference", str(fa), "--bam", str(smoothed),
                        "--sfs", str(sfs)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    vcf = r.stdout
    called = []
    for line in vcf.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        kv = dict(x.split("=", 1) for x in f[7].split(";") if "=" in x)
        called.append((int(f[1]), kv["SVTYPE"], abs(int(kv["SVLEN"])), int(kv["WEIGHT"])))
    truth = [(s.pos, s.kind, s.length) for s in svs]
    assert 5 <= len(called) <= len(truth)
    for pos, kind, length, w in called:
        assert w >= 2 and any(k == kind and abs(l - length) <= 2 and abs(p - pos) <= 15 for p, k, l in truth), (pos, kind, length)
    ref_names, ref_lens, alns = bamio.read_bam(str(smoothed))
    mirror_vcf, info = caller.call(alns, sfs.read_text(), {"chr10M": synth.to_ascii(ref[0])}, list(zip(ref_names, ref_lens)),
                                   ref_names, threads=4, min_sv_length=50)
    assert vcf == mirror_vcf



def _sharded_hip_worker(rank, world, port, q):
    import torch.distributed as dist
    from tests.mirror import multi_call
    from tests.test_multi_cpu import _call_inputs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (the ranks share the one GPU of the test box)
    alns, sfs_text, chromosomes, contigs, ref_names, _ = _call_inputs()
    vcf, info = multi_call.call_sharded(alns, sfs_text, chromosomes, contigs, ref_names, threads=4, min_sv_length=50, device=0)
    q.put((rank, vcf, info["sam"]))
    dist.barrier()
    dist.destroy_process_group()


def test_call_sharded_over_two_ranks_with_the_hip_kernels():
    """SURVEY 8(e) for `call`: POA / realignment batches sharded by sub-cluster index over two ranks (gloo, both on
    this box's GPU), rows gathered, tail run once per rank -- byte-identical to the single-rank call."""
    import socket
    import torch.multiprocessing as mp
    from tests.test_multi_cpu import _call_inputs
    alns, sfs_text, chromosomes, contigs, ref_names, n_truth = _call_inputs()
    vcf0, info0 = caller.call(alns, sfs_text, chromosomes, contigs, ref_names, threads=4, min_sv_length=50)
    assert len([l for l in vcf0.splitlines() if not l.startswith("#")]) == n_truth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, vcf, sam in got:
        assert vcf == vcf0 and sam == info0["sam"]


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (python -m torch.distributed.run, one rank per GPU), here with both
    ranks on the box's one GPU: SVDSS_BENCH_BACKEND=gloo (RCCL refuses two ranks on one device; the exchange then goes
    through host staging).  The N > 1 path end to end on hardware: LPT partition of the contigs, per-rank index built in
    HBM and verified, reads from the rank's contigs, the pre-allocated SFS gather (step i's exchange beside step i+1's
    search, every step's exchange complete inside the timed region), max-over-ranks timing, ONE JSON line from rank 0."""
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SVDSS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "chr20", "--reads", "8192"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "reads/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert "gathered on rank 0" in d["config"]["parallelism"]
    assert d["index_verified_rows"] == 2 * (64_444_167 + 1)
