"""Multi-GPU sharding of the `search` and `call` paths.

Reads are independent (PingPong::process_batch, /root/reference/ping_pong.cpp:176-209,
gives each OpenMP worker a disjoint slice), so the path shards with NO data-path
collective: the index is replicated in every GPU's HBM, each rank searches its
slice of the reads.  The only exchange is the final gather of the (assembled)
SFS records to rank 0, which writes the .sfs text (ping_pong.cpp:213-236).
`call` shards its DP batches (POA, realignment) by sub-cluster index and gathers the per-sub-cluster rows before
the global sort / dedup / chain filter: in the product that is `SVDSS call --gpus N` (csrc/call_host.cpp); its Python
mirror over torch.distributed is tests/mirror/multi_call.py.
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) of n_items for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_reads(flat: np.ndarray, offsets: np.ndarray, rank: int, world: int):
    """Slice of a packed read batch for `rank`: (flat, offsets rebased to 0, first read index)."""
    s, e = shard_range(len(offsets) - 1, rank, world)
    o = offsets[s:e + 1]
    return flat[o[0]:o[-1]], (o - o[0]).astype(np.int64), s


class SfsGatherer:
    """The per-step gather of the assembled SFS on rank 0 (`AllGather(counts)` + direct send/recv),
    built for a step that repeats: the sizes travel in one all_gather of 2 int64 per rank, the payloads as three
    point-to-point messages per peer of exactly the bytes there are (counts as int32, starts, lengths -- no packing, no
    padding to the largest shard), every peer on its own xGMI link, into receive buffers rank 0 allocates once and only
    ever grows.  gather() returns at once with a handle; wait() on it gives rank 0 the views (valid until the next
    gather() into the same slot) -- so the exchange of step i runs beside the search of step i+1 (two slots)."""

    def __init__(self, group=None, slots: int = 2):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gloo = dist.get_backend(group) == "gloo"     # (CPU tests / oversubscribed developer runs: host staging)
        self.slots = [dict(bufs={}, stage=None) for _ in range(slots)]
        self.turn = 0
        self._sizes = None

    def _comm(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if self.gloo and t.is_cuda else t

    def _buf(self, slot, key, n, dtype, dev):
        b = slot["bufs"].get(key)
        if b is None or b.numel() < n:
            b = torch.empty(int(n * 1.25) + 1024, dtype=dtype, device=dev)
            slot["bufs"][key] = b
        return b[:n]

    def gather(self, counts: torch.Tensor, qs: torch.Tensor, ln: torch.Tensor):
        """counts: int64[n_reads_local]; qs / ln: int32[total_local] on the rank's device (the library's own HBM
        buffers: they are copied to a slot's staging buffers first, so the caller may search again at once)."""
        slot = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        if slot.get("pending") is not None:
            self.wait(slot["pending"])
        dev = counts.device
        cdev = torch.device("cpu") if self.gloo else dev
        c32 = self._buf(slot, "c", counts.numel(), torch.int32, dev)
        c32.copy_(counts)
        q = self._buf(slot, "q", qs.numel(), torch.int32, dev)
        q.copy_(qs)
        l = self._buf(slot, "l", ln.numel(), torch.int32, dev)
        l.copy_(ln)
        sizes = torch.tensor([counts.numel(), qs.numel()], dtype=torch.int64, device=cdev)
        all_sizes = torch.empty(2 * self.world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(all_sizes, sizes, group=self.group)
        ops, views = [], None
        if self.rank == 0:
            sz = all_sizes.cpu().view(self.world, 2)
            views = [(c32, q, l)]
            for r in range(1, self.world):
                nr, nrec = int(sz[r, 0]), int(sz[r, 1])
                rc = self._buf(slot, ("rc", r), nr, torch.int32, cdev)
                rq = self._buf(slot, ("rq", r), nrec, torch.int32, cdev)
                rl = self._buf(slot, ("rl", r), nrec, torch.int32, cdev)
                views.append((rc, rq, rl))
                for t in (rc, rq, rl):
                    if t.numel():
                        ops.append(dist.P2POp(dist.irecv, t, r, self.group))
        else:
            send = [self._comm(t) for t in (c32, q, l)]
            slot["stage"] = send          # (kept alive until the sends are done)
            for t in send:
                if t.numel():
                    ops.append(dist.P2POp(dist.isend, t, 0, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        h = {"works": works, "views": views, "slot": slot}
        slot["pending"] = h
        return h

    def wait(self, h, concat: bool = True):
        """Completes the exchange of handle h.  Rank 0 gets (counts int64, starts, lengths) of all ranks in rank order
        (concat=False: the per-rank views [(counts int32, starts, lengths)] as they sit in the receive buffers, nothing
        copied); the other ranks None."""
        for w in h["works"]:
            w.wait()
        h["works"] = []
        if h["slot"].get("pending") is h:
            h["slot"]["pending"] = None
        if self.rank != 0:
            return None
        if not concat:
            return h["views"]
        # (under gloo the peers' slots are host tensors while rank 0's own are where its input was: one device for the cat)
        dev = h["views"][-1][0].device
        cs = torch.cat([v[0].to(dev) for v in h["views"]]).to(torch.int64)
        return cs, torch.cat([v[1].to(dev) for v in h["views"]]), torch.cat([v[2].to(dev) for v in h["views"]])

    def flush(self):
        for s in self.slots:
            if s.get("pending") is not None:
                self.wait(s["pending"])


_default_gatherers = {}


def gather_sfs(counts: torch.Tensor, qs: torch.Tensor, ln: torch.Tensor, group=None
               ) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """Gather per-read SFS records of every rank on rank 0, in rank (= read) order (blocking form of SfsGatherer).
    Returns (counts int64, qs int32, len int32) on rank 0 -- on the device the ranks exchange on --, None elsewhere."""
    g = _default_gatherers.get(group)
    if g is None:
        g = _default_gatherers[group] = SfsGatherer(group, slots=1)
    return g.wait(g.gather(counts, qs, ln))
