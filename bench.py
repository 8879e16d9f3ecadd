#!/usr/bin/env python3
"""bench.py -- reads/s through search+call on MI355X (BASELINE.json's metric, on its configuration).

Default workload (config 4 of BASELINE.json / SURVEY 8(d)): 24 contigs with the GRCh38 primary lengths
(3,088,269,832 bp; both strands indexed: 6.18e9 BWT symbols), HiFi-shape reads of 15 kb with 0.5 % errors.
A "step" = one pass of the hot path over one batch of 1,048,576 reads per GPU (a 30x read set is 6,176,540 reads =
5.9 steps), i.e.
  search  svdss_sfs_search_batch_device: ping-pong search kernel + fused per-read assembly + compaction, reads
          already resident in HBM (ping_pong.cpp:4-49, assembler.cpp:34-56);
  call    the DP kernels of `SVDSS call` for the sub-clusters that many reads imply (config 4: ~20,000 SVs per 30x
          read set => 3,395 clusters of 30 sub-reads per step, het ones split in two by split_cluster):
          svdss_poa_consensus_batch (caller.cpp:257-308), svdss_align_global_batch of every consensus against its
          reference window (caller.cpp:332-355), svdss_indel_ratio_batch on adjacent alleles (caller.cpp:456-458).
          The sub-reads are handed over as host buffers (~120 MB per step), as `pcall` hands them to abPOA.
value = reads / wall time of the steps.  In the same run the index is checked row by row against its text
(svdss_index_verify_device: `index_verified_rows`), the first reads of the batch are searched by the CPU oracle and a
sample of the step's sub-clusters goes through the oracle's POA / realignment / ratio (cpu_baseline: search, call and
the combined rate), and the GPU's results for those reads (counts / starts / lengths / extension counts) and
sub-clusters (consensus lengths, alignment scores, CIGAR lengths, ratios) are compared with them: `verified_reads`,
`verified_subclusters`; the bench fails on a mismatch.  Then the binaries run end to end (file- and PCIe-inclusive,
never `value`): `SVDSS search` on a BAM against a chr20-length and against the whole-genome index, `SVDSS call`.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (config 5): the index is replicated (every rank builds its own replica in its HBM, svdss_index_build_device),
the contigs are dealt to the ranks by LPT bin packing and every rank searches reads drawn from ITS contigs (weak
scaling: 1,048,576 reads per rank per step) and calls its share of the clusters; the only collective is the gather of
the assembled SFS on rank 0 (svdss_amd/multi.py SfsGatherer: RCCL send/recv over xGMI of exactly the bytes there are
into pre-allocated buffers, step i's exchange beside step i+1's search), every step's exchange complete inside the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# The step pipeline keeps several HIP streams busy at once (search, and per call thread: POA x2, realignment, ratio).
# The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a
# queue execute in order; give it enough queues before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz: int32 lane-operations per second (same guide)

GRCH38_PRIMARY = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                  138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                  83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
READS_30X_WG = 6_176_540       # 30 x 3,088,269,832 / 15,000
SVS_30X_WG = 20_000


def lpt_partition(lens, world):
    """Longest-processing-time bin packing of the contigs onto the ranks (config 5)."""
    load = [0] * world
    owner = [0] * len(lens)
    for i in sorted(range(len(lens)), key=lambda i: -lens[i]):
        r = min(range(world), key=lambda r: load[r])
        owner[i] = r
        load[r] += lens[i]
    return owner


def simulate_reads_gpu(ref_t, contig_ranges, n_reads, L, err, seed, device, chunk=2048):
    """Seeded HiFi-shape reads on the GPU (torch ops; data plumbing, not the hot path): uniform start inside one of the
    given contigs [(start, length) in ref_t], random strand, errors sub:ins:del = 2:1.5:1.5 (svdss_amd/synth.py)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comp = torch.tensor([0, 4, 3, 2, 1, 5], dtype=torch.uint8, device=device)
    span = L + L // 20 + 64
    starts0 = torch.tensor([s for s, l in contig_ranges if l > span], dtype=torch.int64, device=device)
    room = torch.tensor([l - span for s, l in contig_ranges if l > span], dtype=torch.int64, device=device)
    cum = torch.cumsum(room, 0)
    total_room = int(cum[-1])
    total = n_reads * L
    out = torch.zeros(((total + 15) // 16) * 16 + 16, dtype=torch.uint8, device=device)
    p_sub, p_ins, p_del = err * 0.4, err * 0.3, err * 0.3
    ar_span = torch.arange(span, device=device)
    ar_L = torch.arange(L, device=device)
    for s in range(0, n_reads, chunk):
        B = min(chunk, n_reads - s)
        u0 = (torch.rand((B,), generator=g, device=device, dtype=torch.float64) * total_room).to(torch.int64)
        u0.clamp_(max=total_room - 1)
        ci = torch.searchsorted(cum, u0, right=True)
        start = starts0[ci] + (u0 - (cum[ci] - room[ci]))
        src = ref_t[start[:, None] + ar_span[None, :]]
        u = torch.rand((B, span), generator=g, device=device)
        is_sub = u < p_sub
        is_ins = (u >= p_sub) & (u < p_sub + p_ins)
        is_del = (u >= p_sub + p_ins) & (u < p_sub + p_ins + p_del)
        reps = torch.ones((B, span), dtype=torch.int32, device=device)
        reps[is_ins] = 2
        reps[is_del] = 0
        c = torch.cumsum(reps, dim=1)
        j = ar_L[None, :].expand(B, L).contiguous().to(torch.int32)
        sidx = torch.searchsorted(c, j, right=True).clamp_(max=span - 1)
        base = torch.gather(src, 1, sidx)
        c_excl = torch.gather(c - reps, 1, sidx)
        first = j == c_excl
        shift = torch.randint(1, 4, (B, L), generator=g, device=device, dtype=torch.uint8)
        subbed = ((base - 1 + shift) % 4) + 1
        acgt = (base >= 1) & (base <= 4)
        base = torch.where(torch.gather(is_sub, 1, sidx) & first & acgt, subbed, base)
        rnd = torch.randint(1, 5, (B, L), generator=g, device=device, dtype=torch.uint8)
        base = torch.where(first, base, rnd)
        strand = torch.rand((B,), generator=g, device=device) < 0.5
        rc = comp[base.flip(1).long()]
        base = torch.where(strand[:, None], rc, base)
        out[s * L:(s + B) * L] = base.reshape(-1)
    offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * L
    return out, offsets


class CallWorkload:
    """The call-side DP work one step's reads imply: n_clusters SVs (INS/DEL alternating, length
    U[50, 2000], 300-bp flanks on both sides as the k-mer extension of the SFS leaves them, clusterer.cpp:159-346), 30
    sub-reads per cluster with 0.5 % errors; every other cluster is heterozygous and leaves split_cluster
    (caller.cpp:100-255) as two sub-clusters of 15 (alt / ref allele), the others as one of 30.  Packed once on the
    host in the layout the C-ABI takes; symbols 0..3 (caller.hpp:25-37)."""

    def __init__(self, n_clusters, seed, coverage=30, err=0.005, flank=300):
        rng = np.random.default_rng(seed)
        seqs, sub_sizes, refs, kinds = [], [], [], []
        for c in range(n_clusters):
            ln = int(rng.integers(50, 2001))
            left = rng.integers(0, 4, size=flank).astype(np.uint8)
            right = rng.integers(0, 4, size=flank).astype(np.uint8)
            seg = rng.integers(0, 4, size=ln).astype(np.uint8)
            ins = c % 2 == 0
            ref_w = np.concatenate([left, right]) if ins else np.concatenate([left, seg, right])
            alt = np.concatenate([left, seg, right]) if ins else np.concatenate([left, right])
            het = (c // 2) % 2 == 0
            groups = [(alt, coverage // 2), (ref_w, coverage - coverage // 2)] if het else [(alt, coverage)]
            for tmpl, k in groups:
                for _ in range(k):
                    r = tmpl.copy()
                    e = rng.random(len(r)) < err
                    r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
                    seqs.append(r)
                sub_sizes.append(k)
                refs.append(ref_w)
                kinds.append((ins, tmpl is alt, ln))
        self.n_clusters = n_clusters
        self.n_sub = len(sub_sizes)
        self.kinds = kinds
        self.seq_off = np.zeros(len(seqs) + 1, dtype=np.int64)
        self.seq_off[1:] = np.cumsum([len(s) for s in seqs])
        self.seqs = np.ascontiguousarray(np.concatenate(seqs))
        self.cluster_off = np.zeros(self.n_sub + 1, dtype=np.int64)
        self.cluster_off[1:] = np.cumsum(sub_sizes)
        self.ref_off = np.zeros(self.n_sub + 1, dtype=np.int64)
        self.ref_off[1:] = np.cumsum([len(r) for r in refs])
        self.refs = np.ascontiguousarray(np.concatenate(refs))
        from svdss_amd.calldp import KSW_MAT           # caller.cpp:333-337: 5 x 5, match 1, mismatch -9, N 0
        self.mat = np.ascontiguousarray(KSW_MAT)
        self._poa = C.c_void_p()
        self._aln = C.c_void_p()
        self.last = {}

    def clone(self):
        """The same packed workload with its own batch objects (one per call thread: device buffers and streams are
        per batch object)."""
        import copy
        c = copy.copy(self)
        c._poa = C.c_void_p()
        c._aln = C.c_void_p()
        c.last = {}
        return c

    def tiled(self, g):
        """The sub-clusters of g consecutive steps as ONE batch (the same packed workload g times, its own batch
        objects): what the call side is handed when it takes several steps' clusters at once (--call-group)."""
        import copy
        c = copy.copy(self)
        n_seq, n_sym, n_ref = len(self.seq_off) - 1, int(self.seq_off[-1]), int(self.ref_off[-1])
        c.seqs = np.ascontiguousarray(np.tile(self.seqs, g))
        c.refs = np.ascontiguousarray(np.tile(self.refs, g))
        c.seq_off = np.concatenate([self.seq_off[:-1] + k * n_sym for k in range(g)] + [[g * n_sym]]).astype(np.int64)
        c.cluster_off = np.concatenate([self.cluster_off[:-1] + k * n_seq for k in range(g)] + [[g * n_seq]]).astype(np.int64)
        c.ref_off = np.concatenate([self.ref_off[:-1] + k * n_ref for k in range(g)] + [[g * n_ref]]).astype(np.int64)
        c.n_clusters, c.n_sub, c.kinds = self.n_clusters * g, self.n_sub * g, self.kinds * g
        # (the SAME batch objects as the step's workload: a call thread runs one batch at a time, and a POA batch object
        # holds up to 32 GB of workspace)
        c.last = {}
        return c

    def run(self, lib, check, device):
        """POA -> consensus to the host -> realignment against the reference windows -> ratio of adjacent consensus
        pairs; returns nothing, leaves timings / results in self.last."""
        t0 = time.perf_counter()
        check(lib.svdss_poa_consensus_batch(self.seqs.ctypes.data, self.seq_off.ctypes.data,
                                            self.cluster_off.ctypes.data, self.n_sub, device, C.byref(self._poa)),
              "svdss_poa_consensus_batch")
        cons_len = np.zeros(self.n_sub, dtype=np.int64)
        cons = np.zeros(max(1, lib.svdss_poa_batch_total(self._poa)), dtype=np.uint8)
        check(lib.svdss_poa_batch_fetch(self._poa, cons_len.ctypes.data, cons.ctypes.data), "svdss_poa_batch_fetch")
        t1 = time.perf_counter()
        q_off = np.zeros(self.n_sub + 1, dtype=np.int64)
        q_off[1:] = np.cumsum(cons_len)
        check(lib.svdss_align_global_batch(cons.ctypes.data, q_off.ctypes.data, self.refs.ctypes.data,
                                           self.ref_off.ctypes.data, self.n_sub, 5, self.mat.ctypes.data, 16, 2, 41, 1,
                                           device, C.byref(self._aln)), "svdss_align_global_batch")
        scores = np.zeros(self.n_sub, dtype=np.int32)
        n_cig = np.zeros(self.n_sub, dtype=np.int64)
        cig = np.zeros(max(1, lib.svdss_aln_batch_total_cigar(self._aln)), dtype=np.uint32)
        check(lib.svdss_aln_batch_fetch(self._aln, scores.ctypes.data, n_cig.ctypes.data, cig.ctypes.data),
              "svdss_aln_batch_fetch")
        t2 = time.perf_counter()
        # chain filter: pairs (k, k + 1) of adjacent consensus sequences -- a and b are two views of one buffer
        ratio = np.zeros(self.n_sub - 1, dtype=np.float64)
        a_off = np.ascontiguousarray(q_off[:-1])
        b_off = np.ascontiguousarray(q_off[1:] - q_off[1])
        check(lib.svdss_indel_ratio_batch(cons.ctypes.data, a_off.ctypes.data, cons.ctypes.data + int(q_off[1]),
                                          b_off.ctypes.data, self.n_sub - 1, device, ratio.ctypes.data, None),
              "svdss_indel_ratio_batch")
        t3 = time.perf_counter()
        self.last = {"poa_wall_ms": (t1 - t0) * 1e3, "realign_wall_ms": (t2 - t1) * 1e3, "ratio_wall_ms": (t3 - t2) * 1e3,
                     "poa_kernel_ms": lib.svdss_poa_batch_kernel_ms(self._poa),
                     "poa_cells": lib.svdss_poa_batch_cells(self._poa), "poa_hbm": lib.svdss_poa_batch_hbm(self._poa),
                     "realign_kernel_ms": lib.svdss_aln_batch_kernel_ms(self._aln),
                     "realign_cells": lib.svdss_aln_batch_cells(self._aln),
                     "cons_len": cons_len, "n_cig": n_cig, "cig": cig, "scores": scores, "ratio": ratio}

    def svs_recovered(self, min_len=50):
        """sub-clusters of the alt allele whose CIGAR carries the implanted I/D (length within 2 %)."""
        ok = n_alt = 0
        o = 0
        for k, (ins, is_alt, ln) in enumerate(self.kinds):
            ops = self.last["cig"][o:o + int(self.last["n_cig"][k])]
            o += int(self.last["n_cig"][k])
            if not is_alt:
                continue
            n_alt += 1
            want = 1 if ins else 2
            if any((int(x) & 0xf) == want and abs((int(x) >> 4) - ln) <= max(2, ln // 50) for x in ops):
                ok += 1
        return ok, n_alt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["wg", "chr20", "wg-families"], default="wg",
                    help="wg (default, the metric's configuration): 24 contigs with GRCh38 primary lengths, 1,048,576 "
                         "reads per GPU per step; chr20: one 64,444,167 bp contig, 30x = 128,888 reads per step; "
                         "wg-families: the wg lengths with 45 %% of the bases in copies of 40 repeat families at "
                         "--divergence (a repeat-rich genome instead of iid sequence; reported beside the wg line, "
                         "never as the headline)")
    ap.add_argument("--divergence", type=float, default=0.05, help="wg-families: divergence of the family copies")
    ap.add_argument("--ref-len", type=int, default=0, help="override: single contig of this many bases")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle leg (also skips verification)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end run of the SVDSS binary on a BAM file")
    ap.add_argument("--e2e-reads", type=int, default=1032000, help="reads in the BAM of the end-to-end run")
    ap.add_argument("--no-e2e-wg", action="store_true", help="skip the end-to-end run against the whole-genome index")
    ap.add_argument("--no-e2e-call", action="store_true", help="skip the end-to-end run of `SVDSS call`")
    ap.add_argument("--no-e2e-30x", action="store_true", help="skip the chain of the binaries at the metric's own scale (e2e_chain_30x)")
    ap.add_argument("--chain-reads", type=int, default=READS_30X_WG, help="reads of e2e_chain_30x (default: the 30x set, 6,176,540)")
    ap.add_argument("--no-call-dp", action="store_true", help="search only (value is then NOT the headline metric)")
    ap.add_argument("--no-gather", action="store_true", help="multi-GPU: leave the SFS on the ranks")
    ap.add_argument("--search-threads", type=int, default=1,
                    help="batch objects / threads that take the steps' searches in turn (the launch of step i+1 is under "
                         "way while step i runs its tail kernels)")
    ap.add_argument("--call-threads", type=int, default=3,
                    help="steps whose call-side DP may be in flight at once (0: no pipelining, search and call of a "
                         "step run back to back)")
    ap.add_argument("--call-group", type=int, default=1,
                    help="the call side takes the sub-clusters of this many consecutive steps as one batch (`SVDSS call` "
                         "is handed a whole genome's at once: a single step's 5,093 leave the chip waiting for their "
                         "longest chains); every step's sub-clusters are processed inside the timed region either way; "
                         "steps that do not fill a group go one by one; 1 (default) = a batch per step.  Measured: the step does not "
                         "move with it (profiles/r05x_call_group.txt) -- in the pipeline the search fills the tails already")
    args = ap.parse_args()
    args.call_group = max(1, args.call_group)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # SVDSS_BENCH_BACKEND=gloo: developer check of the N>1 code path on a box with fewer GPUs than ranks (the ranks
    # then share GPUs; RCCL refuses that).  The driver's runs use the default, RCCL, one rank per GPU.
    backend = os.environ.get("SVDSS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import svdss_amd
    from svdss_amd import multi, synth
    from svdss_amd._lib import check, lib

    if args.ref_len:
        contig_lens = [args.ref_len]
    elif args.workload in ("wg", "wg-families"):
        contig_lens = GRCH38_PRIMARY
    else:
        contig_lens = [64_444_167]
    ref_total = sum(contig_lens)
    L = args.read_len
    if args.reads:
        n_reads = args.reads
    elif args.workload in ("wg", "wg-families") and not args.ref_len:
        n_reads = 1 << 20
    else:
        n_reads = int(round(args.coverage * ref_total / L))
    gather = world > 1 and not args.no_gather

    # ---- index: every rank builds its own replica in its HBM (svdss_index_build_device) -------
    t0 = time.time()
    families = args.workload == "wg-families"
    ref = (synth.make_family_reference(contig_lens, seed=11, divergence=args.divergence) if families
           else synth.make_reference(contig_lens, seed=11))
    t_ref = time.time() - t0
    t0 = time.time()
    ix = svdss_amd.FMDIndex.build(ref, device=local_rank)
    t_index = time.time() - t0
    # the index against its own text, row by row, by code that shares nothing with the builder (csrc/index_verify.hip):
    # the oracle below is built from this index's BWT and must not inherit an unchecked index
    t0 = time.time()
    iv = ix.verify()
    t_verify = time.time() - t0
    if iv["rows"] != ix.size or iv["first_bad"] != -1 or any(iv[k] for k in ("bad_order", "bad_bwt", "bad_range", "bad_block", "bad_dollar")):
        raise SystemExit(f"INDEX VERIFICATION FAILED: {iv}")
    e2e_dir, e2e_error = None, None
    if rank == 0 and world == 1 and not args.no_e2e:
        # (the end-to-end legs are auxiliary: whatever goes wrong there -- a full /tmp, a missing binary -- is reported
        # in the JSON line and must not cost the run its headline measurement)
        import shutil
        import tempfile
        e2e_error = None
        try:
            e2e_dir = tempfile.mkdtemp(prefix="svdss_bench_e2e_", dir="/tmp")
            e2e_prepare(e2e_dir, ref, wg=(len(contig_lens) == 24 and not args.no_e2e_wg and not families))
        except Exception as e:   # noqa: BLE001
            e2e_error = f"prepare: {type(e).__name__}: {e}"
            if e2e_dir:
                shutil.rmtree(e2e_dir, ignore_errors=True)
            e2e_dir = None

    # ---- reads: generated on the GPU from the contigs LPT gives this rank -------
    owner = lpt_partition(contig_lens, world)
    starts = np.concatenate([[0], np.cumsum(contig_lens)[:-1]])
    mine = [(int(starts[i]), int(contig_lens[i])) for i in range(len(contig_lens)) if owner[i] == rank]
    if len(contig_lens) < world:      # fewer contigs than ranks (chr20 workload): equal slices of the reference instead
        sl = ref_total // world
        mine = [(rank * sl, sl)]
    ref_t = torch.from_numpy(ref[0] if len(ref) == 1 else np.concatenate(ref)).to(device)
    del ref
    d_reads, d_offs = simulate_reads_gpu(ref_t, mine, n_reads, L, args.err, seed=13 + 1000 * rank, device=device)
    total_syms = n_reads * L
    del ref_t
    torch.cuda.synchronize()
    # the read simulator's temporaries sit in torch's caching allocator: hand them back, the call side sizes its POA
    # workspace (a few MB per sub-cluster) by what the device has free
    torch.cuda.empty_cache()
    hbm_free_after_reads = torch.cuda.mem_get_info(device)[0]

    # ---- call-side work of one step's reads ------
    cw = None
    if not args.no_call_dp:
        n_clusters = max(2, int(round(SVS_30X_WG * n_reads / READS_30X_WG)))   # the SV density per read of config 4
        cw = CallWorkload(n_clusters, seed=99 + rank)

    pp = svdss_amd.PingPong(ix, assemble=True)
    # the search runs on its own non-blocking stream (the library's call-side entry points use private streams too), so
    # that the call-side DP of earlier steps overlaps the search of the next one
    def search_stream():
        # (SVDSS_SEARCH_CUS / SVDSS_CALL_CUS: the search and the call-side DP on disjoint compute units -- the stream
        # then comes from the library, which knows how to restrict it)
        if os.environ.get("SVDSS_SEARCH_CUS"):
            import ctypes as C
            h = C.c_void_p()
            check(lib.svdss_search_stream_create(local_rank, C.byref(h)), "svdss_search_stream_create")
            return torch.cuda.ExternalStream(h.value, device=device)
        return torch.cuda.Stream(device=device)

    sstream = search_stream()

    gatherer = multi.SfsGatherer(slots=2) if gather else None
    pending_gather = []

    def search(assemble=True, pp=pp, sstream=sstream):
        pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), n_reads, total_syms,
                                   stream=sstream.cuda_stream, assemble=assemble, fetch=False)
        if gather and assemble:
            # the exchange of this step is posted (results copied to a staging slot first) and waited for when the
            # NEXT step's exchange is posted: it runs beside the next search
            with torch.cuda.stream(sstream):
                counts, qs, ln = pp.device_results()
                h = gatherer.gather(counts, qs, ln)
                while pending_gather:
                    gatherer.wait(pending_gather.pop(), concat=False)
                pending_gather.append(h)

    def finish_gathers():
        if gather:
            with torch.cuda.stream(sstream):
                while pending_gather:
                    gatherer.wait(pending_gather.pop(), concat=False)
            sstream.synchronize()

    # --search-threads S > 1: S batch objects / streams / threads take the steps' searches in turn, so that the launch of
    # step i+1 is under way while step i runs its short tail kernels and host-side waits (order, scan, gather)
    searchers = [(pp, sstream)] + [(svdss_amd.PingPong(ix, assemble=True), search_stream())
                                   for _ in range(max(0, (1 if gather else args.search_threads) - 1))]

    stats = {"kernel_ms": [], "pipeline_ms": [], "poa_ms": [], "aln_ms": [], "call_wall_ms": [], "call_steps": []}

    def run_steps(n_steps, record):
        """n_steps steps: step i = search(i), then call(i).  With --call-threads T > 0 the steps are software-pipelined:
        one thread searches step after step, T threads take the call-side DP of the steps whose search is done -- call(i)
        overlaps search(i+1..) and the calls of neighbouring steps (a sub-cluster is one long chain of dependent row
        steps: a single step's POA batch leaves most of the chip idle while its longest chains finish)."""
        import queue
        import threading
        G = args.call_group
        n_grouped = (n_steps // G) * G if G > 1 else 0     # steps whose sub-clusters go G steps at a time
        if cw is None or args.call_threads <= 0:
            for i in range(n_steps):
                search()
                if record:
                    stats["kernel_ms"].append(pp.last_search_kernel_ms)
                    stats["pipeline_ms"].append(pp.last_kernel_ms)
                if cw is not None:
                    if i < n_grouped:
                        if (i + 1) % G == 0:
                            workers[0][1].run(lib, check, local_rank)
                            if record:
                                note_call(workers[0][1], G)
                    else:
                        cw.run(lib, check, local_rank)
                        if record:
                            note_call(cw, 1)
            return
        todo = queue.Queue()
        errors = []

        def caller(w):
            torch.cuda.set_device(local_rank)
            try:
                while True:
                    n = todo.get()
                    if n is None:
                        return
                    b = w[1] if n > 1 else w[0]      # the batch of G steps' sub-clusters / of one step's
                    b.run(lib, check, local_rank)
                    if record:
                        note_call(b, n)
            except BaseException as e:   # noqa: BLE001  (re-raised in the main thread)
                errors.append(e)

        threads = [threading.Thread(target=caller, args=(w,)) for w in workers]
        for t in threads:
            t.start()
        steps_q = queue.Queue()
        for i in range(n_steps):
            steps_q.put(i)
        searched = [0]

        def searcher(spp, sst):
            torch.cuda.set_device(local_rank)
            try:
                while True:
                    try:
                        i = steps_q.get_nowait()
                    except queue.Empty:
                        return
                    search(pp=spp, sstream=sst)
                    if record:
                        with lock:
                            stats["kernel_ms"].append(spp.last_search_kernel_ms)   # HIP events on the search stream
                            stats["pipeline_ms"].append(spp.last_kernel_ms)
                    with lock:       # the call side is handed the steps whose search is done, G at a time
                        searched[0] += 1
                        k = searched[0]
                    if k > n_grouped:
                        todo.put(1)
                    elif k % G == 0:
                        todo.put(G)
            except BaseException as e:   # noqa: BLE001
                errors.append(e)

        sthreads = [threading.Thread(target=searcher, args=sp) for sp in searchers[1:]]
        try:
            for t in sthreads:
                t.start()
            searcher(*searchers[0])
            for t in sthreads:
                t.join()
        finally:
            for _ in threads:
                todo.put(None)
            for t in threads:
                t.join()
        if errors:
            raise errors[0]

    lock = __import__("threading").Lock()

    def note_call(w, n):
        with lock:
            stats["poa_ms"].append(w.last["poa_kernel_ms"])
            stats["aln_ms"].append(w.last["realign_kernel_ms"])
            stats["call_wall_ms"].append(w.last["poa_wall_ms"] + w.last["realign_wall_ms"] + w.last["ratio_wall_ms"])
            stats["call_steps"].append(n)

    # per call thread: a batch object for one step's sub-clusters and one for a group's
    workers = []
    if cw is not None:
        for t in range(max(1, args.call_threads)):
            one = cw if t == 0 else cw.clone()
            workers.append((one, one.tiled(args.call_group) if args.call_group > 1 else None))
    gw = workers[0][1] if workers else None

    # raw (unassembled) SFS count: the N_sfs of SURVEY 8(d)'s reference-model bytes; and the search kernel on an
    # otherwise idle GPU
    search(assemble=False)
    n_sfs_raw = pp.last_total
    alone_ms = []
    for _ in range(2):
        search()
        alone_ms.append(pp.last_search_kernel_ms)
    call_alone = None
    if cw is not None:
        for w in workers:       # first call of every batch object: allocates its device arena
            for b in w:
                if b is not None:
                    b.run(lib, check, local_rank)
        cw.run(lib, check, local_rank)
        call_alone = dict(cw.last)
        group_alone = None
        if gw is not None:
            gw.run(lib, check, local_rank)
            group_alone = dict(gw.last)
            # the group is the step's workload G times: G times the step's results
            for key in ("cons_len", "n_cig", "scores"):
                if not np.array_equal(gw.last[key], np.tile(cw.last[key], args.call_group)):
                    raise SystemExit(f"CALL GROUP MISMATCH: {key} of the batch of {args.call_group} steps is not {args.call_group} x the step's")
            if not np.array_equal(gw.last["cig"], np.tile(cw.last["cig"], args.call_group)):
                raise SystemExit("CALL GROUP MISMATCH: CIGARs")
    run_steps(args.warmup, record=False)
    finish_gathers()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, record=True)
    finish_gathers()          # every step's exchange completes inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, pipeline_ms = stats["kernel_ms"], stats["pipeline_ms"]
    poa_ms, aln_ms, call_wall_ms = stats["poa_ms"], stats["aln_ms"], stats["call_wall_ms"]
    search(assemble=True)     # leave the assembled results of one more search in the batch object for the verification
    finish_gathers()

    n_ext = pp.last_total_ext
    n_sfs_asm = pp.last_total
    k_ms = float(np.mean(kernel_ms))

    if rank == 0:
        wl = "GRCh38 primary lengths" if len(contig_lens) == 24 else "chr20 length" if ref_total == 64_444_167 else "custom"
        out = {
            "metric": ("reads/sec through search+call, 30x HiFi 15 kb reads vs 3 Gb ref, 1/2/4/8 GPU"
                       + (" [repeat-rich variant of the reference: NOT the headline configuration]" if families else ""))
                      if cw is not None else "reads/sec through SFS search only (NOT the headline metric: --no-call-dp)",
            "value": world * n_reads * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic {ref_total} bp reference in {len(contig_lens)} contig(s) ({wl}, "
                             + (f"45% of the bases in copies of 40 repeat families at {args.divergence * 100:g}% divergence"
                                if families else "iid ACGT + 3% diverged repeats")
                             + f", both strands indexed: {ix.size} BWT symbols), {n_reads} reads/GPU/step x {L} bp "
                             f"({n_reads * L / ref_total:.2f}x per step; a 30x set = {30 * ref_total / L / n_reads:.2f} steps), "
                             f"{args.err * 100:.2f}% errors; step = search (ping-pong + fused assemble, all reads searched = "
                             "--noputative semantics, reads resident in HBM)"
                             + (f" + call DP for the {cw.n_clusters} clusters / {cw.n_sub} sub-clusters those reads imply "
                                f"(20,000 SVs per 30x: POA of {int(cw.cluster_off[-1])} sub-reads, realignment, chain-filter "
                                "ratio; sub-reads handed over as host buffers)" if cw is not None else "")),
                "reads_per_gpu": n_reads, "read_len": L, "index_bytes": ix.device_bytes, "hbm_free_before_the_steps": hbm_free_after_reads,
                "parallelism": (f"contigs dealt to {world} GPU(s) by LPT, reads drawn from each rank's contigs, index "
                                "replicated (built per rank in HBM), "
                                + ("assembled SFS gathered on rank 0 over RCCL every step" if gather
                                   else "no data-path collective")),
                "ext_per_read": n_ext / n_reads, "raw_sfs_per_read": n_sfs_raw / n_reads,
                "assembled_sfs_per_read": n_sfs_asm / n_reads, "reference_build_s": round(t_ref, 1),
                "index_build_s": round(t_index, 1), "index_verify_s": round(t_verify, 2), "kmer_table_k": ix.kmer_k, "segments_per_read": pp.last_segments,
                "reads_redone_unsegmented": pp.last_fallbacks,
                "search_ms_per_step": float(np.mean(pipeline_ms)),
                "search_kernel_ms_on_idle_gpu": float(np.mean(alone_ms)),
                "pipelining": (f"{args.call_threads} call thread(s): the call-side DP of up to {args.call_threads} earlier "
                               f"call batches (of {args.call_group} step(s) each) runs beside the search of the current step" if cw is not None and args.call_threads > 0
                               else "none: search and call of a step back to back"),
            },
        }
        oracle_fm = None
        out["index_verified_rows"] = iv["rows"]
        out["roofline"] = search_roofline(ref_total, n_reads, L, ix.kmer_k, k_ms, n_ext, total_syms, n_sfs_raw,
                                          float(np.mean(pipeline_ms)), float(np.mean(alone_ms)),
                                          tag=(f"_families{args.divergence:g}" if families else ""))
        if cw is not None:
            okc, n_alt = cw.svs_recovered()
            # per STEP: the batches of the timed region covered call_steps steps each
            n_call_steps = float(sum(stats["call_steps"]))
            poa_k, aln_k = float(np.sum(poa_ms)) / n_call_steps, float(np.sum(aln_ms)) / n_call_steps
            out["config"]["call_dp"] = {
                "clusters": cw.n_clusters, "subclusters": cw.n_sub, "subreads": int(cw.cluster_off[-1]),
                "call_group_steps": args.call_group,
                "call_batches_in_the_timed_region": {str(g): stats["call_steps"].count(g) for g in sorted(set(stats["call_steps"]))},
                "group_call_on_idle_gpu": ({"subclusters": gw.n_sub, "poa_kernel_ms": round(group_alone["poa_kernel_ms"], 3),
                                            "realign_kernel_ms": round(group_alone["realign_kernel_ms"], 3),
                                            "poa_gcups": group_alone["poa_cells"] / (group_alone["poa_kernel_ms"] * 1e-3) / 1e9,
                                            "wall_ms": round(group_alone["poa_wall_ms"] + group_alone["realign_wall_ms"] + group_alone["ratio_wall_ms"], 3)}
                                           if group_alone is not None else None),
                "call_wall_ms_per_step": float(np.sum(call_wall_ms)) / n_call_steps,
                "one_call_on_idle_gpu": {"poa_kernel_ms": round(call_alone["poa_kernel_ms"], 3),
                                         "realign_kernel_ms": round(call_alone["realign_kernel_ms"], 3),
                                         "ratio_wall_ms": round(call_alone["ratio_wall_ms"], 3),
                                         "wall_ms": round(call_alone["poa_wall_ms"] + call_alone["realign_wall_ms"]
                                                          + call_alone["ratio_wall_ms"], 3)},
                "poa_kernel_ms": round(poa_k, 3), "poa_cells": cw.last["poa_cells"],
                "poa_gcups": cw.last["poa_cells"] / (poa_k * 1e-3) / 1e9, "poa_subclusters_on_hbm_kernel": cw.last["poa_hbm"],
                "realign_kernel_ms": round(aln_k, 3), "realign_cells": cw.last["realign_cells"],
                "realign_gcups": cw.last["realign_cells"] / (aln_k * 1e-3) / 1e9,
                "ratio_wall_ms": round(cw.last["ratio_wall_ms"], 3),
                "alt_subclusters_with_the_implanted_sv_in_the_cigar": f"{okc}/{n_alt}",
                # integer DP, not HBM- and not MFMA-bound: cell updates x VALU lane-operations per cell
                # against the int32 issue rate of 256 CUs x 4 SIMD-32 x 2.4 GHz
            }
            # (hoisted out of config.call_dp, which the driver's record truncates: the call-side kernels against their bound,
            # measured alone on an idle GPU; poa_kernel_ms / realign_kernel_ms in call_dp are per step inside the pipeline)
            out["roofline"]["call_side"] = call_rooflines(cw.last["poa_cells"], call_alone["poa_kernel_ms"], cw.last["realign_cells"],
                                                          call_alone["realign_kernel_ms"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["verified_reads"], out["verified_subclusters"], oracle_fm = cpu_baseline_and_verify(
                ix, pp, d_reads, L, n_reads, args.cpu_seconds, cw)
        if e2e_error:
            out["e2e_error"] = e2e_error
        if world == 1 and not args.no_e2e and e2e_dir:
            # the end-to-end run is a process of its own on the same GPU: this one lets go of the index (127 GB), the
            # reads and the call-side arenas first
            for sp, _ in searchers:
                sp.close()
            for w, _g in workers:      # (a pair shares its batch objects)
                for h, free in ((w._poa, lib.svdss_poa_batch_free), (w._aln, lib.svdss_aln_batch_free)):
                    if h:
                        free(h)
                w._poa.value = None
                w._aln.value = None
            ix.close()
            del d_reads, d_offs
            torch.cuda.empty_cache()
            time.sleep(6.0)   # (the driver clears the ~190 GB this process just handed back; see _run_search)
            try:
                out.update(e2e_runs(e2e_dir, args.e2e_reads, call=not args.no_e2e_call,
                                    chain_reads=0 if args.no_e2e_30x else args.chain_reads, oracle_fm=oracle_fm,
                                    cpu_call=(out.get("cpu_baseline") or {}).get("call")))
                if "e2e_chain_30x" in out:
                    c30 = out["e2e_chain_30x"]
                    out["binaries"] = {
                        "what": "files in, files out: SVDSS smooth -> search -> call on the metric's own configuration (e2e_chain_30x); "
                                "`value` above is the kernels on reads resident in HBM",
                        "search_plus_call_reads_per_s": c30["search_plus_call_reads_per_s"], "chain_reads_per_s": c30["chain_reads_per_s"],
                        "reads": c30["reads"], "coverage": c30["coverage"]}
            except Exception as e:   # noqa: BLE001
                out["e2e_error"] = f"{type(e).__name__}: {str(e)[-400:]}"
            finally:
                import shutil
                shutil.rmtree(e2e_dir, ignore_errors=True)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# VALU lane-operations per DP cell, counted in the kernels' inner loops:
POA_OPS_PER_CELL = 30      # convex-gap POA cell: 2 predecessors x (H, E1, E2) max/add + F scan share + direction word
ALN_OPS_PER_CELL = 24      # dual-affine cell: H, E, F, E2, F2 + direction byte


def call_rooflines(poa_cells, poa_ms, aln_cells, aln_ms):
    def one(cells, ms, ops):
        ach = cells / (ms * 1e-3) / 1e9
        peak = VALU_LANE_OPS / ops / 1e9
        return {"bound": "valu-int32", "achieved": ach, "peak": peak, "unit": "GCUPS", "frac": ach / peak,
                "lane_ops_per_cell": ops}
    return {"poa": one(poa_cells, poa_ms, POA_OPS_PER_CELL), "realign": one(aln_cells, aln_ms, ALN_OPS_PER_CELL)}


def search_roofline(ref_total, n_reads, L, k, k_ms, n_ext, total_syms, n_sfs_raw, all_ms, alone_ms, tag=""):
    """HBM roofline of the search kernel.  `achieved` = the bytes the kernel's own memory operations move per launch
    (TCC_EA0_RDREQ x 128 B + WRREQ x 32|64 B from the committed rocprofv3 --pmc pass of exactly this workload,
    profiles/traffic.json) / the kernel time measured live with HIP events on the launch stream; null when this
    workload has not been profiled.  The reference algorithm's cost model (one 64-B block per
    rb3_fmd_extend) is reported as `reference_model_bytes`: the kernel answers most of those extensions from its k-mer
    table / text compare without fetching their blocks, so that figure is not a bound of this kernel."""
    prof = profiled(ref_total, n_reads, L, k, tag)
    ref_model = n_ext * 64 + total_syms + 16 * n_sfs_raw
    r = {"bound": "hbm", "kernel": "sfs_search2_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": k_ms,
         "all_search_kernels_ms": all_ms, "kernel_ms_on_idle_gpu": alone_ms, "reference_model_bytes": ref_model,
         "reference_model_gbs": ref_model / (k_ms * 1e-3) / 1e9}
    if prof:
        traffic = prof["read_requests"] * 128 + prof["write_bytes"]
        r.update({"achieved": traffic / (k_ms * 1e-3) / 1e9, "traffic": traffic, "traffic_source": prof["source"],
                  "kernel_hash": prof["kernel_hash"],
                  "useful_bytes": prof.get("useful_bytes"),
                  "lines_per_read": prof["read_requests"] / n_reads})
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
        # (kernel_ms is the average over the timed region, where the call-side kernels of earlier steps share the chip)
        r["frac_on_idle_gpu"] = traffic / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if prof.get("useful_bytes"):
            r["traffic_over_useful"] = traffic / prof["useful_bytes"]
        probe = random_probe()
        if probe:
            rate = prof["read_requests"] / (k_ms * 1e-3)
            r["random_access"] = {"read_lines_per_s": rate,
                                  "ceiling_independent_lines_per_s": probe["independent_random_lines_per_s"],
                                  "frac_of_ceiling": rate / probe["independent_random_lines_per_s"],
                                  "source": probe["source"]}
    else:
        r.update({"achieved": None, "traffic": None, "frac": None,
                  "note": "no committed --pmc pass for this workload and this version of the kernel (hash "
                          + search_kernel_hash() + " of " + " ".join(SEARCH_KERNEL_SOURCES) + "): profiles/traffic.json"})
    return r


SEARCH_KERNEL_SOURCES = ("sfs_search.hip", "sfs_core2.h", "fmd_layout.h", "sym_window.h")


def search_kernel_hash():
    """sha256 (16 hex digits) over the sources the search kernel is compiled from: a committed counter pass
    (profiles/traffic.json) is only quoted for the kernel it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in SEARCH_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "svdss_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profiled(ref_total, n_reads, L, k, tag=""):
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            table = json.load(fh)
    except OSError:
        return None
    e = table.get(f"ref{ref_total}_reads{n_reads}_len{L}_k{k}{tag}")
    if not isinstance(e, dict) or e.get("kernel_hash") != search_kernel_hash():
        return None          # never profiled, or profiled on another version of the kernel: no fraction is claimed
    return e


def random_probe():
    try:
        with open(os.path.join(ROOT, "profiles", "random_access.json")) as fh:
            return json.load(fh)
    except OSError:
        return None


def cpu_quota():
    """processors this container may use: visible hardware threads capped by the cgroup's CPU quota"""
    n = os.cpu_count() or 1
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


E2E_CHR_BP = 64_444_167


def e2e_prepare(work, ref, wg):
    """Files of the end-to-end runs, written while the host copy of the reference exists: chr.fa = the first
    64,444,167 bases of the first contig, wg.fa = all contigs (the whole-genome run), ref0.npy = chr.fa's bases 0..3 for
    the BAM writer."""
    lut = np.frombuffer(b"NACGTN", dtype=np.uint8)
    first = ref[0][:E2E_CHR_BP]
    with open(os.path.join(work, "chr.fa"), "wb") as f:
        f.write(b">chrS\n")
        f.write(lut[first].tobytes())
        f.write(b"\n")
    np.save(os.path.join(work, "ref0.npy"), (first - 1) & 3)
    if wg:
        with open(os.path.join(work, "wg.fa"), "wb") as f:
            for i, c in enumerate(ref):
                f.write(b">c%d\n" % (i + 1))
                f.write(lut[c].tobytes())
                f.write(b"\n")


def _run_search(exe, fmd, bam, env=None, repeats=1, pause_s=0.0):
    """`SVDSS search --bam` -> dict of timings from its --verbose log.  repeats > 1: that many runs of the process, the
    one with the MEDIAN streaming time reported and every run's streaming seconds listed beside it (`streaming_s_runs`):
    the second run over a freshly generated input file waits most of a second for the file's loaders (the page cache, not
    this code: runs 1, 3, 4, 5 do not, nor does any run a second apart; profiles/r04y_consecutive_runs.txt)."""
    if repeats > 1:
        # pause_s: seconds between the end of one process and the start of the next.  A process that ends hands ~130 GB of
        # HBM back (whole-genome index) and the driver clears them at 30-50 GB/s: a process started right behind it waits
        # for clean memory in its first allocations (profiles/r05z_e2e_lib_ab.txt).  Back-to-back runs are this bench's
        # doing, not a user's.
        runs = []
        for _ in range(repeats):
            if pause_s > 0:
                time.sleep(pause_s)
            runs.append(_run_search(exe, fmd, bam, env))
        runs_sorted = sorted(runs, key=lambda r: r["streaming_s"])
        out = dict(runs_sorted[len(runs) // 2])
        out["streaming_s_runs"] = [r["streaming_s"] for r in runs]
        out["streaming_s_min_median_max"] = [runs_sorted[0]["streaming_s"], out["streaming_s"], runs_sorted[-1]["streaming_s"]]
        return out
    import re
    import subprocess
    t0 = time.perf_counter()
    r = subprocess.run([exe, "search", "--index", fmd, "--bam", bam, "--noputative", "--verbose"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, check=True,
                       env=dict(os.environ, SVDSS_DEBUG="1", **(env or {})))
    wall = time.perf_counter() - t0
    t_file = float(re.search(r"index file read at \+([0-9.]+) s", r.stderr).group(1))
    t_ix = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1))
    m = re.search(r"(\d+) records read, (\d+) SFS written at \+([0-9.]+) s", r.stderr)
    n, n_sfs, t_end = int(m.group(1)), int(m.group(2)), float(m.group(3))
    g = re.search(r"(\d+) chunks inflated on the GPU", r.stderr)
    c = re.search(r"\] (\d+) chunks: locate", r.stderr)
    d = re.search(r"device path: (\d+) batches, (\d+) segments \((\d+) walked again\); busy seconds: GPU batches ([0-9.]+) \(inflate kernels ([0-9.]+)\)", r.stderr)
    out = {"reads": n, "sfs": n_sfs, "streaming_s": round(t_end - t_ix, 3), "index_file_read_s": round(t_file, 3),
           "index_restore_s": round(t_ix, 3), "end_of_output_s": round(t_end, 3), "whole_process_s": round(wall, 3),
           "reads_per_s_streaming": n / max(t_end - t_ix, 1e-9), "whole_process_reads_per_s": n / wall}
    e = re.search(r"front end beside the index restore: (\d+) batches \((\d+) records\) .* their (\d+) reads searched in (\d+) launch\(es\), ([0-9.]+) s", r.stderr)
    if e:
        out["front_end_beside_restore"] = {"batches_parked": int(e.group(1)), "records_read_before_the_index_was_resident": int(e.group(2)),
                                           "reads_parked": int(e.group(3)), "launches": int(e.group(4)), "parked_search_s": float(e.group(5))}
    k = re.search(r"table of order (\d+): few reads to search", r.stderr)
    if k:
        out["kmer_table_order"] = int(k.group(1))
    if "rank blocks alone" in r.stderr:
        out["index_mode"] = "the rank blocks alone (few reads to search): no text, suffix array or k-mer table"
    if d:   # records handled on the GPU (csrc/bam_device.hip): only compressed bytes went up
        out.update({"path": "device (BAM records walked, filtered and unpacked on the GPU)", "device_batches": int(d.group(1)),
                    "record_chain_segments": int(d.group(2)), "segments_walked_again": int(d.group(3)),
                    "inflate_kernel_s_summed": float(d.group(5))})
    else:
        out.update({"path": "host (records sliced on the host)", "bgzf_chunks": int(c.group(1)) if c else None,
                    "bgzf_chunks_inflated_on_gpu": int(g.group(1)) if g else 0})
    return out


def _search_leg(exe, fmd, bam, pause_s=5.0):
    """One `SVDSS search --bam` leg: (i) the STREAMING rate, index first and then the file (SVDSS_SEARCH_EARLY=0: the order of
    ping_pong.cpp:245,329; three runs, the median one reported, min / median / max beside it) -- the rate a long input
    approaches; (ii) the binary as it runs by default since round 6, the BAM front end beside the index restore (reads
    parked in HBM, searched in large launches when the index is resident): whole-process seconds of two runs."""
    r = _run_search(exe, fmd, bam, env={"SVDSS_SEARCH_EARLY": "0"}, repeats=3, pause_s=pause_s)
    r["whole_process_s_index_first"] = r.pop("whole_process_s")
    r["whole_process_reads_per_s_index_first"] = r.pop("whole_process_reads_per_s")
    d = []
    for _ in range(2):
        time.sleep(pause_s)
        d.append(_run_search(exe, fmd, bam))
    d.sort(key=lambda x: x["whole_process_s"])
    r["whole_process_s"] = d[0]["whole_process_s"]
    r["whole_process_s_runs"] = [x["whole_process_s"] for x in d]
    r["whole_process_reads_per_s"] = d[0]["whole_process_reads_per_s"]
    r["default_run"] = {k: d[0].get(k) for k in ("index_restore_s", "end_of_output_s", "whole_process_s", "front_end_beside_restore", "kmer_table_order", "index_mode")}
    r["streaming_is"] = "SVDSS_SEARCH_EARLY=0 (index first, then the file); whole_process_s is the default run (front end beside the index restore)"
    return r


def e2e_runs(work, n_reads, call=True, chain_reads=0, oracle_fm=None, cpu_call=None):
    """The binaries as a user of run_svdss sees them (file- and PCIe-inclusive, never `value`), on the GPU the main
    measurement has just let go of.
      e2e       `SVDSS search --bam`: a synthetic BAM of 15 kb reads against a chr20-length index -- BGZF inflate (GPU),
                record parsing, 4-bit upload, search, text to /dev/null.  e2e_reads_per_s = reads / (end of output -
                index resident), the streaming rate a 30x sample (6.2 M reads) approaches.
      e2e_wg    the same BAM against the index of the WHOLE reference (GRCh38 primary lengths): what run_svdss:151-166
                does for a human sample; `SVDSS index` time and the restore time (records file read + index rebuilt in
                HBM + k-mer table) are stated beside the streaming rate.
      e2e_call  `SVDSS index` -> `search` -> `call` on a 100 Mb genome with 648 implanted SVs (config 4's density), 30x of
                15 kb error-free ("smoothed") reads with truth alignments: call_reads_per_s = reads / wall of `call`,
                search_plus_call_reads_per_s = reads / (wall of search + wall of call) (run_svdss:151-178)."""
    import subprocess
    from tools import e2e_search as E
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    out = {}
    unit = 172000            # (the record blocks of `unit` reads are written n_reads / unit times)
    repeat = max(1, round(n_reads / unit))
    ref0 = np.load(os.path.join(work, "ref0.npy"))
    bam = os.path.join(work, "reads.bam")
    raw = E.write_bam(bam, "chrS", ref0, unit, 15000, repeat=repeat)
    os.sync()   # (15 GB of freshly written pages.  It does not cure what it was added for: the SECOND run over a new file
                #  waits 0.4-0.9 s for the file's loaders, with or without it -- tools/r04_backtoback.sh -- hence the medians)
    del ref0
    t0 = time.perf_counter()
    subprocess.run([exe, "index", "-d", os.path.join(work, "chr.fa"), "-o", os.path.join(work, "chr.fmd")], check=True, capture_output=True)
    t_index = time.perf_counter() - t0
    r = _search_leg(exe, os.path.join(work, "chr.fmd"), bam, pause_s=3.0)
    out["e2e_reads_per_s"] = r["reads_per_s_streaming"]
    r["what"] = ("SVDSS search --bam (binary): synthetic BAM, %d x 15 kb reads, %.1f GB file (%.1f GB inflated), "
                 "chr20-length index, text to /dev/null; the run with the median streaming time of three, 5 s apart (min / median / max beside it)" % (r["reads"], os.path.getsize(bam) / 1e9, raw / 1e9))
    r["index_s"] = round(t_index, 2)
    r["host_cpu_quota_cores"] = cpu_quota()
    out["e2e"] = r
    try:
        # the same run with the records sliced on the host (the path of rounds 1-3, kept as the fallback)
        h = _run_search(exe, os.path.join(work, "chr.fmd"), bam, env={"SVDSS_BAM_DEVICE": "0"})
        out["e2e_host_path"] = {k: h[k] for k in ("path", "streaming_s", "reads_per_s_streaming", "whole_process_s")}
    except Exception as e:   # noqa: BLE001
        out["e2e_host_path_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
    try:
        # run_svdss:151-165: `SVDSS smooth` writes the BAM that `SVDSS search` reads.  smooth's output is deflated by this
        # repository's own encoder (csrc/deflate.hip: dynamic Huffman, literals only), which the GPU inflater decodes
        # about twice as fast as zlib's level-1 streams with their thousands of 3-byte matches per block
        sm = os.path.join(work, "smoothed.bam")
        t0 = time.perf_counter()
        with open(sm, "wb") as f:
            subprocess.run([exe, "smooth", "--reference", os.path.join(work, "chr.fa"), "--bam", bam, "--threads", "16"], check=True,
                           stdout=f, stderr=subprocess.DEVNULL)
        t_smooth = time.perf_counter() - t0
        r2 = _search_leg(exe, os.path.join(work, "chr.fmd"), sm, pause_s=3.0)
        r2["what"] = ("SVDSS smooth -> SVDSS search (binaries), as run_svdss:151-165 chains them: search reads the BAM smooth wrote "
                      "(%.1f GB, deflated on the GPU); smooth: %.2f s whole process = %.0f reads/s"
                      % (os.path.getsize(sm) / 1e9, t_smooth, r["reads"] / t_smooth))
        r2["smooth_s"] = round(t_smooth, 3)
        r2["smooth_reads_per_s"] = r["reads"] / t_smooth
        out["e2e_smoothed"] = r2
        os.remove(sm)
    except Exception as e:   # noqa: BLE001
        out["e2e_smoothed_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
    if os.path.exists(os.path.join(work, "wg.fa")):
        try:
            t0 = time.perf_counter()
            subprocess.run([exe, "index", "-d", os.path.join(work, "wg.fa"), "-o", os.path.join(work, "wg.fmd")], check=True, capture_output=True)
            t_index = time.perf_counter() - t0
            r = _search_leg(exe, os.path.join(work, "wg.fmd"), bam, pause_s=5.0)
            r["what"] = ("the same BAM against the index of the whole reference (24 contigs, GRCh38 primary lengths, 6.18e9 BWT "
                         "symbols), every run 5 s after the process before ended (the driver clears the HBM a process hands back; "
                         "back to back the next one streams beside that): restore = records file (%.1f GB) read + index rebuilt in HBM + K = 16 table"
                         % (os.path.getsize(os.path.join(work, "wg.fmd.svdss")) / 1e9))
            r["index_s"] = round(t_index, 2)
            r["fmd_bytes"] = os.path.getsize(os.path.join(work, "wg.fmd"))
            out["e2e_wg"] = r
        except Exception as e:   # noqa: BLE001
            out["e2e_wg_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
    os.remove(bam)
    if call:
        try:
            from tools import e2e_call as EC
            cdir = os.path.join(work, "call")
            fa, cbam, svs, het, recs, hdr, _ = EC.write_dataset(cdir, 100_000_000, 648, 30, 15000, het_every=2, max_len=2000)
            fmd = os.path.join(cdir, "ref.fmd")
            subprocess.run([exe, "index", "-d", fa, "-o", fmd], check=True, capture_output=True)
            sfs = os.path.join(cdir, "specifics.txt")
            t0 = time.perf_counter()
            with open(sfs, "wb") as f:
                subprocess.run([exe, "search", "--index", fmd, "--bam", cbam], check=True, stdout=f, stderr=subprocess.DEVNULL)
            t_search = time.perf_counter() - t0
            t0 = time.perf_counter()
            vcf = subprocess.run([exe, "call", "--reference", fa, "--bam", cbam, "--sfs", sfs, "--threads", "16",
                                  "--min-sv-length", "50"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            t_call = time.perf_counter() - t0
            called = [c[:3] for c in EC.parse_vcf(vcf)]
            truth = [(sv.pos, sv.kind, sv.length) for sv in svs]
            hit = sum(1 for p, k, l in truth if any(k == ck and l == cl and abs(cp - p) <= 12 for cp, ck, cl in called))
            n = len(recs)
            out["e2e_call"] = {"what": "SVDSS index -> search -> call (binaries): 100 Mb genome, %d implanted SVs (every other one "
                                       "heterozygous), %d error-free 15 kb reads with truth alignments (30x), whole-process wall "
                                       "times" % (len(svs), n),
                               "reads": n, "search_s": round(t_search, 3), "call_s": round(t_call, 3),
                               "call_reads_per_s": n / t_call, "search_plus_call_reads_per_s": n / (t_search + t_call),
                               "svs_called": len(called), "svs_truth": len(truth), "truth_recovered": hit}
        except Exception as e:   # noqa: BLE001
            out["e2e_call_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
        shutil_rm = __import__("shutil").rmtree
        shutil_rm(os.path.join(work, "call"), ignore_errors=True)
        have_wg = os.path.exists(os.path.join(work, "wg.fa")) and os.path.exists(os.path.join(work, "wg.fmd"))
        from tools import e2e_call_wg as W

        # The generated INPUT of a chain (reads.bam + .bai: 20 GB at 30x) goes to /dev/shm when that has room: the driver's box
        # has a 79 GB /tmp, and with the reference, the index and both BAMs on it the file system is 81 % full while `smooth`
        # writes -- its writers then fall behind (7.5 - 7.9 s instead of 5.2 - 5.4 s alone; profiles/r06w_*).  Either way the
        # input is read from memory (the generator has just written it: page cache or tmpfs); outputs stay on /tmp.
        shm_dir = None
        try:
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize > (64 << 30) and not os.environ.get("SVDSS_BENCH_NO_SHM"):
                shm_dir = os.path.join("/dev/shm", "svdss_bench_%d" % os.getpid())
                os.makedirs(shm_dir, exist_ok=True)
        except OSError:
            shm_dir = None

        def chain_dir():
            # the chains run on THIS run's reference (wg.fa, the contigs the headline index was built from) and on the index
            # `SVDSS index` made of it for e2e_wg: ref.fa / ref.fmd are links, tools/chain_dataset.cpp reads the FASTA
            d = os.path.join(work, "chainwg")
            os.makedirs(d, exist_ok=True)
            for link, target in (("ref.fa", "wg.fa"), ("ref.fmd", "wg.fmd"), ("ref.fmd.svdss", "wg.fmd.svdss")):
                if have_wg and not os.path.lexists(os.path.join(d, link)):
                    os.symlink(os.path.join(work, target), os.path.join(d, link))
            if shm_dir:
                for f in ("reads.bam", "reads.bam.bai"):
                    for q in (os.path.join(shm_dir, f), os.path.join(d, f)):
                        if os.path.lexists(q):
                            os.remove(q)
                    os.symlink(os.path.join(shm_dir, f), os.path.join(d, f))      # (the generator writes through the link)
            return d

        try:
            # the chain a user of run_svdss runs (run_svdss:136-178) on 1.03 M reads (5x), and as its last two
            # stages what `e2e_call_wg` has reported since round 4.  Round 6: the reads carry errors sub:ins:del = 2:1.5:1.5
            # with the indels in their CIGARs (tools/chain_dataset.cpp; rounds 4-5: substitutions only)
            d = chain_dir()
            r = W.run_chain(d, 1_030_000, 3400, ref_from_fasta=have_wg)
            r["what"] = ("SVDSS index -> smooth -> search (putative) -> call (binaries) at whole-genome lengths: 24 contigs with the GRCh38 "
                         "primary lengths, 3,400 implanted SVs (every other one heterozygous: at 5x many of those stay below "
                         "--min-cluster-weight), 1,030,000 x 15 kb reads (5x) with 0.5 % errors sub:ins:del = 2:1.5:1.5 and truth alignments; "
                         "smooth rewrites them to the reference, search reads the smoothed BAM and skips what smooth tagged XF != 0, call "
                         "reads the ORIGINAL BAM through its .bai (run_svdss:167-176); whole-process wall times, the index restore of search included")
            out["e2e_chain_wg"] = r
            out["e2e_call_wg"] = {k: r.get(k) for k in ("reads", "svs", "reference_bp", "index_s", "search_s", "search_index_resident_s", "call_s",
                                                        "svs_called", "truth_recovered", "search_plus_call_reads_per_s", "call_log", "search_log")}
            out["e2e_call_wg"]["call_reads_per_s"] = r["reads"] / r["call_s"]
            out["e2e_call_wg"]["what"] = "the search and call stages of e2e_chain_wg (since round 5 the reads reach them through SVDSS smooth)"
            for f in ("reads.bam", "reads.bam.bai", "smoothed.bam", "specifics.txt"):
                if os.path.exists(os.path.join(d, f)):
                    os.remove(os.path.join(d, f))
        except Exception as e:   # noqa: BLE001
            out["e2e_chain_wg_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
        if chain_reads > 0:
            try:
                # THE METRIC'S OWN CONFIGURATION through the binaries (BASELINE.json config 4): 30x = 6,176,540
                # reads, 20,000 SVs, run_svdss's order.  ~50 GB of /tmp while it runs.
                d = chain_dir()
                n_svs = max(1, round(SVS_30X_WG * chain_reads / READS_30X_WG))
                r = W.run_chain(d, chain_reads, n_svs, ref_from_fasta=have_wg, keep=True)
                r["what"] = ("BASELINE.json config 4 through the binaries, as run_svdss:136-178 chains them: 24 contigs with the GRCh38 primary "
                             "lengths, %d x 15 kb reads (%.1fx) with 0.5 %% errors sub:ins:del = 2:1.5:1.5 and truth alignments in a sorted, "
                             "indexed BAM, %d implanted SVs (INS / DEL, 50-2000 bp, every other pair heterozygous); SVDSS smooth -> SVDSS search "
                             "(putative, on the smoothed BAM) -> SVDSS call (original BAM + .bai); whole-process wall times of each binary, "
                             "files in and out (page cache warm: the generator has just written them), search includes its index restore"
                             % (r["reads"], r["coverage"], r["svs"]))
                try:
                    r["cpu_baseline"] = cpu_chain_baseline(d, r, oracle_fm, cpu_call)
                except Exception as e:   # noqa: BLE001
                    r["cpu_baseline_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
                r["input_bam_on"] = ("/dev/shm (tmpfs): the box's /tmp is too small to hold the reference, the index and both BAMs without slowing "
                                     "smooth's writers down (profiles/r06w_*)") if shm_dir else "the work directory (/tmp)"
                out["e2e_chain_30x"] = r
            except Exception as e:   # noqa: BLE001
                out["e2e_chain_30x_error"] = f"{type(e).__name__}: {str(e)[-300:]}"
        shutil_rm(os.path.join(work, "chainwg"), ignore_errors=True)
        if shm_dir:
            shutil_rm(shm_dir, ignore_errors=True)
    return out


def cpu_chain_baseline(work, chain, fm, cpu_call, max_file_bytes=256 << 20, max_reads=4096):
    """The CPU cost of the SAME chain on the SAME files, from measured pieces on a bounded sample (VERDICT r5 missing #5):
      * BGZF: zlib inflate of the first members of the chain's BAMs and zlib level-6 deflate of what they hold (what htslib
        does around every stage), single-thread rates scaled to the files' sizes and divided by the cores;
      * search: the reads `SVDSS smooth` tagged XF:0 among the first records of the smoothed BAM -- the ones `search`
        (putative) looks at -- through the oracle's ping-pong search + assemble on all cores, scaled to the XF:0 reads of the file;
      * call: the oracle's POA + realignment + ratio rate on the step's sub-clusters (cpu_baseline.call), times the clusters
        `call` reported.
    Smoothing's CIGAR walk, placement and clustering are left out (cheap next to these): the figure is a LOWER bound of the
    CPU time, the GPU / CPU ratio conservative.  Test infrastructure timed as a baseline; never on the product path."""
    import re
    import struct
    import zlib
    from tests import oracle_lib as O
    threads = min(O.max_threads(), cpu_quota())
    sm = os.path.join(work, "smoothed.bam")
    data = open(sm, "rb").read(max_file_bytes)
    # members
    t_inf, raw_parts, comp_bytes, o = 0.0, [], 0, 0
    while o + 18 <= len(data):
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        if o + bsize > len(data):
            break
        t0 = time.perf_counter()
        raw_parts.append(zlib.decompress(data[o + 18:o + bsize - 8], -15))
        t_inf += time.perf_counter() - t0
        comp_bytes += bsize
        o += bsize
    raw = b"".join(raw_parts)
    inflate_bps = len(raw) / max(t_inf, 1e-9)
    t0 = time.perf_counter()
    defl_sample = raw[:64 << 20]
    zc = zlib.compressobj(6, zlib.DEFLATED, -15)
    zc.compress(defl_sample)
    zc.flush()
    deflate_bps = len(defl_sample) / max(time.perf_counter() - t0, 1e-9)
    # records: XF:0 reads -> nt6
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 4 + struct.unpack_from("<i", raw, p)[0] + 4
    lut = np.full(16, 5, dtype=np.uint8)
    lut[[1, 2, 4, 8]] = [1, 2, 3, 4]
    reads, n_rec, p_first = [], 0, p
    while p + 4 <= len(raw) and len(reads) < max_reads:
        bs = struct.unpack_from("<i", raw, p)[0]
        if p + 4 + bs > len(raw):
            break
        l_name, n_cig, l_seq = raw[p + 12], struct.unpack_from("<H", raw, p + 16)[0], struct.unpack_from("<i", raw, p + 20)[0]
        q = p + 36 + l_name + 4 * n_cig
        aux = q + (l_seq + 1) // 2 + l_seq
        end = p + 4 + bs
        xf = None
        while aux + 3 <= end:      # (smooth writes XF as a one-byte integer; other tags of these files: none)
            tag, ty = raw[aux:aux + 2], chr(raw[aux + 2])
            size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}.get(ty)
            if size is None:
                break
            if tag == b"XF":
                xf = int.from_bytes(raw[aux + 3:aux + 3 + size], "little")
            aux += 3 + size
        n_rec += 1
        if xf == 0:
            packed = np.frombuffer(raw, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q)
            b = np.empty(2 * len(packed), dtype=np.uint8)
            b[0::2] = lut[packed >> 4]
            b[1::2] = lut[packed & 15]
            reads.append(b[:l_seq])
        p = end
    if fm is None or not reads:
        raise RuntimeError("no oracle index / no XF:0 reads in the sample")
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(x) for x in reads])
    flat = np.ascontiguousarray(np.concatenate(reads))
    t0 = time.perf_counter()
    c, _, _, _ = fm.search_batch(flat, offs, True, threads)
    t_search = time.perf_counter() - t0
    search_rate = len(reads) / t_search
    m = re.search(r"XF 0/1/2/3: (\d+) ", " ".join(chain.get("smooth_log", [])))
    n_xf0 = int(m.group(1)) if m else int(chain["reads"] * len(reads) / max(n_rec, 1))
    n_sub = max(chain.get("svs_called", 0), chain.get("svs", 0))
    sub_rate = (cpu_call or {}).get("value")
    # inflated sizes: ~ reads x (32 + name + cigar + 7.5 kb of bases + 15 kb of qualities)
    inflated_per_read = (p - p_first) / max(n_rec, 1)
    inflated = chain["reads"] * inflated_per_read
    smooth_s = (inflated / inflate_bps + inflated / deflate_bps) / threads
    search_s = inflated / inflate_bps / threads + n_xf0 / search_rate
    call_s = 1.3 * inflated / inflate_bps / threads + (n_sub / sub_rate if sub_rate else 0.0)
    total = smooth_s + search_s + call_s
    return {"kind": "port", "cores": threads, "unit": "reads/s",
            "chain_reads_per_s": chain["reads"] / total, "search_plus_call_reads_per_s": chain["reads"] / (search_s + call_s),
            "seconds": {"smooth": round(smooth_s, 1), "search": round(search_s, 1), "call": round(call_s, 1)},
            "measured": {"zlib_inflate_MBps_per_core": round(inflate_bps / 1e6, 1), "zlib6_deflate_MBps_per_core": round(deflate_bps / 1e6, 1),
                         "oracle_search_reads_per_s": round(search_rate, 1), "oracle_call_subclusters_per_s": sub_rate,
                         "sample_records": n_rec, "sample_xf0_reads": len(reads), "sample_sfs": int(c.sum()), "search_sample_s": round(t_search, 2)},
            "scaled_to": {"reads": chain["reads"], "xf0_reads": n_xf0, "subclusters": n_sub, "inflated_bytes_per_bam": int(inflated)},
            "sample": (f"first {comp_bytes >> 20} MB of the chain's smoothed BAM: zlib inflate / level-6 deflate single-thread rates; its "
                       f"{len(reads)} XF:0 reads through oracle/svdss_oracle.c on {threads} threads; call DP at cpu_baseline.call's rate; "
                       "per-stage seconds = BGZF bytes / rate / cores + searched reads / rate + sub-clusters / rate (smoothing's CIGAR "
                       "walk, placement, clustering not counted: a lower bound of the CPU time)")}


def cpu_baseline_and_verify(ix, pp, d_reads, L, n_reads, target_s, cw):
    """The oracle timed on this box's cores on a bounded sample of the same work -- and, since it computes those
    results anyway, the checker of the GPU's.
      search  CPU restatement of ping_pong.cpp:4-49 + assembler.cpp:34-56, OpenMP over reads like ping_pong.cpp:329, on
              the first reads of the batch; the GPU's counts / starts / lengths / extension counts must equal it;
      call    oracle POA (caller.cpp:257-308), extd2 realignment (:332-355) and chain-filter ratio (:456-458), OpenMP over
              sub-clusters like caller.cpp:319-321, on the first sub-clusters of the step; the GPU's consensus lengths,
              alignment scores, CIGAR lengths and ratios must equal it.
    value = reads/s of search + call together: the reads of a step / (their search time + the call time of the
    sub-clusters those reads imply), each scaled from its sample."""
    from tests import oracle_lib as O
    fm = O.OracleFMD.from_bwt(ix.bwt())
    threads = min(O.max_threads(), cpu_quota())     # (more threads than the container's CPU quota only add contention)

    def run(k):
        flat = d_reads[:k * L].cpu().numpy()
        offs = np.arange(k + 1, dtype=np.int64) * L
        t0 = time.perf_counter()
        res = fm.search_batch(flat, offs, True, threads)
        return time.perf_counter() - t0, res

    k = min(n_reads, max(4 * threads, 512))     # (a first sample large enough to predict the second one)
    t, _ = run(k)
    k2 = int(min(n_reads, max(k, k * target_s / max(t, 1e-3))))
    t2, (c, q, l, e) = run(k2)
    got = pp._fetch()      # results of the last timed step (assembled)
    tot = int(c.sum())
    ok = ((got.counts[:k2] == c).all() and (got.n_ext[:k2] == e).all() and int(got.counts[:k2].sum()) == tot
          and (got.qs[:tot] == q).all() and (got.len[:tot] == l).all())
    if not ok:
        raise SystemExit(f"VERIFICATION FAILED: GPU SFS of the first {k2} reads differ from the oracle's")
    search_rate = k2 / t2
    base = {"value": search_rate, "unit": "reads/s", "cores": threads, "kind": "port",
            "search": {"value": search_rate, "unit": "reads/s", "sample_reads": k2, "seconds": round(t2, 2)},
            "sample": f"search: first {k2} reads of the rank-0 batch, {t2:.1f} s, oracle/svdss_oracle.c orc_search_batch with "
                      f"{threads} OpenMP threads (search + assemble; plain sampled-Occ FMD, faster than ropebwt3's "
                      "rld0: the GPU/CPU ratio is conservative)"}
    m2 = 0
    if cw is not None:
        def run_call(m):
            so, co, ro = cw.seq_off, cw.cluster_off, cw.ref_off
            t0 = time.perf_counter()
            res = O.call_batch(cw.seqs[:so[co[m]]], so[:co[m] + 1], co[:m + 1], cw.refs[:ro[m]], ro[:m + 1], cw.mat, threads)
            return time.perf_counter() - t0, res
        m = min(cw.n_sub, 4 * threads)
        t, _ = run_call(m)
        m2 = int(min(cw.n_sub, max(m, m * (0.7 * target_s) / max(t, 1e-3))))
        t2c, (cl, sc, nc, ra) = run_call(m2)
        g = cw.last
        okc = ((g["cons_len"][:m2] == cl).all() and (g["scores"][:m2] == sc).all() and (g["n_cig"][:m2] == nc).all()
               and (g["ratio"][:m2 - 1] == ra).all())
        if not okc:
            raise SystemExit(f"VERIFICATION FAILED: GPU consensus / alignment / ratio of the first {m2} sub-clusters differ "
                             "from the oracle's")
        sub_rate = m2 / t2c
        # a step's reads imply cw.n_sub sub-clusters
        call_s_per_step = cw.n_sub / sub_rate
        search_s_per_step = n_reads / search_rate
        base["call"] = {"value": sub_rate, "unit": "sub-clusters/s", "sample_subclusters": m2, "seconds": round(t2c, 2),
                        "reads_equivalent_per_s": n_reads / call_s_per_step}
        base["value"] = n_reads / (search_s_per_step + call_s_per_step)
        base["sample"] += (f"; call: first {m2} of the step's {cw.n_sub} sub-clusters, {t2c:.1f} s, oracle POA + extd2 + ratio "
                           f"(oracle/svdss_oracle_callbatch.c) with {threads} OpenMP threads; value = reads of a step / (their "
                           "search time + the call time of the sub-clusters they imply)")
    return base, k2, m2, fm


if __name__ == "__main__":
    main()
