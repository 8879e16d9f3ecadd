"""N>1 path on CPU: world_size-2 gloo run of the sharding + SFS gather used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from svdss_amd import multi


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


@pytest.mark.parametrize("world", [2, 3])
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
