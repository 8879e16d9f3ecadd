"""Multi-GPU sharding of the `search` and `call` paths.

Reads are independent (PingPong::process_batch, /root/reference/ping_pong.cpp:176-209,
gives each OpenMP worker a disjoint slice), so the path shards with NO data-path
collective: the index is replicated in every GPU's HBM, each rank searches its
slice of the reads.  The only exchange is the final gather of the (assembled)
SFS records to rank 0, which writes the .sfs text (ping_pong.cpp:213-236).
`call` shards its DP batches (POA, realignment) by sub-cluster index and gathers the
per-sub-cluster rows before the global sort / dedup / chain filter (call_sharded).
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


def gather_sfs(counts: torch.Tensor, qs: torch.Tensor, ln: torch.Tensor, group=None
               ) -> Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """Gather per-read SFS records of every rank on rank 0, in rank (= read) order.

    counts: int64[n_reads_local]; qs/ln: int32[total_local], all on the rank's device.
    Sizes are exchanged with one all_gather of 2 int64 per rank, payloads with one
    gather of buffers padded to the largest shard (RCCL needs equal sizes; at <=
    a few hundred MB per node the padding is cheaper than a second round of sizes).
    Returns (counts, qs, len) on rank 0, None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = counts.device
    sizes = torch.tensor([counts.numel(), qs.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu()
    max_reads = int(all_sizes[:, 0].max())
    max_recs = int(all_sizes[:, 1].max())
    # one int64 payload per rank: [counts | qs<<32|len]
    payload = torch.zeros(max_reads + max_recs, dtype=torch.int64, device=dev)
    payload[:counts.numel()] = counts
    if qs.numel():
        payload[max_reads:max_reads + qs.numel()] = (qs.to(torch.int64) << 32) | ln.to(torch.int64)
    # direct gather: every peer sends to rank 0 on its own xGMI link (RCCL implements
    # gather as grouped send/recv), nothing is replicated to ranks that do not need it
    bufs = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, gather_list=bufs, dst=0, group=group)
    if rank != 0:
        return None
    cs, qq, ll = [], [], []
    for r in range(world):
        nr, nrec = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        cs.append(bufs[r][:nr])
        rec = bufs[r][max_reads:max_reads + nrec]
        qq.append((rec >> 32).to(torch.int32))
        ll.append((rec & 0xFFFFFFFF).to(torch.int32))
    return torch.cat(cs), torch.cat(qq), torch.cat(ll)



def call_sharded(alignments, sfs_text: str, chromosomes: dict, contigs, ref_names, group=None, **kw):
    """`SVDSS call` over the ranks of one node (one process per GPU, torch.distributed already initialised):
    svdss_amd.caller.call with the POA / realignment batches sharded by sub-cluster index and one
    all_gather_object of the per-sub-cluster rows (consensus, score, CIGAR -- a few hundred bytes each).  Every rank
    returns the same (vcf_text, info), byte-identical to the single-GPU call.  kw: caller.call's keyword arguments
    (device defaults to this rank's current CUDA device)."""
    from . import caller
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(rows):
        parts = [None] * world
        dist.all_gather_object(parts, rows, group=group)
        return parts

    if "device" not in kw and torch.cuda.is_available():
        kw["device"] = torch.cuda.current_device()
    return caller.call(alignments, sfs_text, chromosomes, contigs, ref_names, shard=(rank, world, all_gather), **kw)
