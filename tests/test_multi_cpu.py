"""N>1 path on CPU: world_size-2 gloo run of the sharding + SFS gather used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from svdss_amd import multi
from tests.mirror import multi_call


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 128890):
        for w in (1, 2, 4, 8):
            rs = [multi.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [e - s for s, e in rs]
            assert max(sizes) - min(sizes) <= 1


def test_shard_reads_rebases_offsets():
    offs = np.array([0, 3, 3, 10, 12], dtype=np.int64)
    flat = np.arange(12, dtype=np.uint8)
    got = [multi.shard_reads(flat, offs, r, 2) for r in range(2)]
    assert got[0][1].tolist() == [0, 3, 3] and got[0][0].tolist() == [0, 1, 2] and got[0][2] == 0
    assert got[1][1].tolist() == [0, 7, 9] and got[1][0].tolist() == list(range(3, 12)) and got[1][2] == 2


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n_reads = 5 + 3 * rank                      # ragged shards, one rank may have no SFS at all
    counts = rng.integers(0, 4, size=n_reads) if rank != 1 else np.zeros(n_reads, dtype=np.int64)
    total = int(counts.sum())
    qs = rng.integers(0, 15000, size=total).astype(np.int32)
    ln = rng.integers(1, 2000, size=total).astype(np.int32)
    res = multi.gather_sfs(torch.from_numpy(counts.astype(np.int64)), torch.from_numpy(qs), torch.from_numpy(ln))
    if rank == 0:
        q.put([t.numpy() for t in res])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_steps(rank, world, port, q):
    """Three steps through one two-slot SfsGatherer: step i's exchange is waited for only after step i+1 was posted;
    the receive buffers are reused (grow-only) and every step's result is exact."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = multi.SfsGatherer(slots=2)
    out, prev = [], None
    for step in range(3):
        rng = np.random.default_rng(1000 * step + rank)
        n_reads = 4 + 2 * rank + step
        counts = rng.integers(0, 5 + 3 * step, size=n_reads)
        total = int(counts.sum())
        qs = rng.integers(0, 15000, size=total).astype(np.int32)
        ln = rng.integers(1, 2000, size=total).astype(np.int32)
        h = g.gather(torch.from_numpy(counts.astype(np.int64)), torch.from_numpy(qs), torch.from_numpy(ln))
        if prev is not None:
            r = g.wait(prev)
            if rank == 0:
                out.append([t.numpy().copy() for t in r])
        prev = h
    r = g.wait(prev)
    if rank == 0:
        out.append([t.numpy().copy() for t in r])
        q.put(out)
    g.flush()
    dist.barrier()
    dist.destroy_process_group()


def test_gatherer_overlapped_steps_gloo():
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_steps, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for step in range(3):
        cs, qq, ll = [], [], []
        for rank in range(world):
            rng = np.random.default_rng(1000 * step + rank)
            counts = rng.integers(0, 5 + 3 * step, size=4 + 2 * rank + step)
            total = int(counts.sum())
            cs.append(counts)
            qq.append(rng.integers(0, 15000, size=total).astype(np.int32))
            ll.append(rng.integers(1, 2000, size=total).astype(np.int32))
        c, qv, lv = got[step]
        assert (c == np.concatenate(cs)).all() and (qv == np.concatenate(qq)).all() and (lv == np.concatenate(ll)).all()


@pytest.mark.parametrize("world", [2, 3, 8])      # (8: the node the metric is quoted on)
def test_gather_sfs_gloo(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    counts, qs, ln = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp_c, exp_q, exp_l = [], [], []
    for rank in range(world):
        rng = np.random.default_rng(100 + rank)
        n_reads = 5 + 3 * rank
        c = rng.integers(0, 4, size=n_reads) if rank != 1 else np.zeros(n_reads, dtype=np.int64)
        t = int(c.sum())
        exp_c.append(c.astype(np.int64))
        exp_q.append(rng.integers(0, 15000, size=t).astype(np.int32))
        exp_l.append(rng.integers(1, 2000, size=t).astype(np.int32))
    assert (counts == np.concatenate(exp_c)).all()
    assert (qs == np.concatenate(exp_q)).all() and (ln == np.concatenate(exp_l)).all()


# ---- `call` over N ranks: DP batches sharded by sub-cluster index, one gather of the rows ---------
# The DP functions are replaced by the CPU oracle here (this test is about the sharding and the gather; the HIP
# kernels behind the default functions are compared with the same oracle in the -m gpu tests).

def _oracle_fns():
    from tests.mirror import caller
    from tests import oracle_lib as O

    def poa_fn(clusters, device=0):
        return ["".join("ACGTN"[x] for x in O.poa_consensus([caller.encode26(s) for s in cl])) for cl in clusters], {}

    def align_fn(queries, targets, device=0):
        res = [O.ksw_extd2_global(caller.encode26(q), caller.encode26(t), caller.KSW_MAT) for q, t in zip(queries, targets)]
        return np.array([r[0] for r in res], dtype=np.int32), [r[1] for r in res], {}

    def ratio_fn(a_list, b_list, device=0):
        return np.array([O.fuzz_ratio(a.encode(), b.encode()) for a, b in zip(a_list, b_list)]), None

    return dict(poa_fn=poa_fn, align_fn=align_fn, ratio_fn=ratio_fn)


def _call_inputs():
    import tempfile
    import svdss_amd
    from svdss_amd import pingpong, synth
    from tests.mirror import bamio
    from tests import bam_writer
    from tests import oracle_lib as O
    from tests.pipeline_sim import simulate
    ref, svs, reads = simulate(ref_lens=(60000, 40000), n_svs=6, coverage=10, read_len=3000, seed=21)
    names = ["chrA", "chrB"]
    recs = [bam_writer.record(n, 0, tid, pos, 60, cig, seq, [("XF", "C", 0)]) for n, tid, pos, cig, seq, hp in reads]
    with tempfile.NamedTemporaryFile(suffix=".bam") as fh:
        fh.write(bam_writer.bam([(n, len(c)) for n, c in zip(names, ref)], recs))
        fh.flush()
        ref_names, ref_lens, alns = bamio.read_bam(fh.name)
    fm = O.OracleFMD.build(ref)
    flat, offs = svdss_amd.pack_reads([pingpong.nt6_encode(r[4]) for r in reads])
    c, q, l, _ = fm.search_batch(flat, offs, True)
    sols, o = [], 0
    for r, n in zip(reads, c.tolist()):
        if n:
            sols.append((r[0], 0, list(zip(q[o:o + n].tolist(), l[o:o + n].tolist()))))
        o += n
    sfs_text = pingpong.output_batch(sols)
    chromosomes = {n: synth.to_ascii(cc) for n, cc in zip(names, ref)}
    return alns, sfs_text, chromosomes, list(zip(names, ref_lens)), ref_names, len(svs)


def _call_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    alns, sfs_text, chromosomes, contigs, ref_names, _ = _call_inputs()
    vcf, info = multi_call.call_sharded(alns, sfs_text, chromosomes, contigs, ref_names, threads=4, min_sv_length=50,
                                   device=0, **_oracle_fns())
    q.put((rank, vcf, info["sam"], info["subclusters"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_call_sharded_gloo_equals_single_rank(world):
    from tests.mirror import caller
    alns, sfs_text, chromosomes, contigs, ref_names, n_truth = _call_inputs()
    vcf0, info0 = caller.call(alns, sfs_text, chromosomes, contigs, ref_names, threads=4, min_sv_length=50, **_oracle_fns())
    rows0 = [l for l in vcf0.splitlines() if not l.startswith("#")]
    assert len(rows0) == n_truth and info0["subclusters"] >= n_truth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_call_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == list(range(world))
    for _, vcf, sam, n_sub in got:          # every rank: the single-rank bytes
        assert vcf == vcf0 and sam == info0["sam"] and n_sub == info0["subclusters"]
