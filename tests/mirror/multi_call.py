"""`SVDSS call` over the ranks of one node, on the Python mirror of the host logic (tests/mirror/caller.py): the
sharding rule of csrc/call_host.cpp (sub-cluster k -> shard k % N, one global sort / dedup / chain filter) checked with
world-2/3 gloo runs on the CPU and a 2-rank HIP run on the GPU box."""
import torch
import torch.distributed as dist


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
