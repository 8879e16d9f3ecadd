#!/usr/bin/env python3
"""bench.py -- reads/s through the SFS-search hot path on MI355X.

A "step" = one pass of the hot path (svdss_sfs_search_batch_device: ping-pong
search kernel + fused per-read assembly + compaction) over one batch of
synthetic HiFi-shape reads that is already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: reads shard across ranks (weak scaling: every rank gets its own
batch of the same size), the index is replicated in every GPU's HBM, and the
path has no exchange step: every rank keeps the SFS of its shard (as N
`SVDSS search` processes would each write their part of the .sfs file).
`--gather` adds a gather of the assembled SFS on rank 0 (svdss_amd/multi.py,
RCCL send/recv over xGMI) to every step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

GRCH38_PRIMARY = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                  138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                  83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def simulate_reads_gpu(ref_t, n_reads, L, err, seed, device, chunk=2048):
    """Seeded HiFi-shape reads on the GPU (torch ops; data plumbing, not the hot path):
    uniform start, random strand, errors sub:ins:del = 2:1.5:1.5 (svdss_amd/synth.py semantics)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    comp = torch.tensor([0, 4, 3, 2, 1, 5], dtype=torch.uint8, device=device)
    span = L + L // 20 + 64
    n_ref = ref_t.numel()
    total = n_reads * L
    out = torch.zeros(((total + 15) // 16) * 16 + 16, dtype=torch.uint8, device=device)
    p_sub, p_ins, p_del = err * 0.4, err * 0.3, err * 0.3
    ar_span = torch.arange(span, device=device)
    ar_L = torch.arange(L, device=device)
    for s in range(0, n_reads, chunk):
        B = min(chunk, n_reads - s)
        start = torch.randint(0, n_ref - span, (B,), generator=g, device=device)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["chr20", "wg"], default="chr20",
                    help="chr20: one 64,444,167 bp contig, 30x; wg: 24 contigs with GRCh38 primary lengths "
                         "(3,088,269,832 bp), 1,048,576 reads per step (a 30x set is 6.2 M reads = 6 steps)")
    ap.add_argument("--ref-len", type=int, default=0, help="override: single contig of this many bases")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: coverage*ref/read_len)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-call-dp", action="store_true", help="skip the call-side DP kernel measurement")
    ap.add_argument("--gather", action="store_true", help="multi-GPU: gather the SFS on rank 0 inside every timed step")
    args = ap.parse_args()

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

    if args.ref_len:
        contig_lens = [args.ref_len]
    elif args.workload == "wg":
        contig_lens = GRCH38_PRIMARY
    else:
        contig_lens = [64_444_167]
    ref_total = sum(contig_lens)
    if args.reads:
        n_reads = args.reads
    elif args.workload == "wg" and not args.ref_len:
        n_reads = 1 << 20
    else:
        n_reads = int(round(args.coverage * ref_total / args.read_len))
    L = args.read_len

    # ---- index: built once (rank 0), replicated into every GPU's HBM -------
    t0 = time.time()
    ref = synth.make_reference(contig_lens, seed=11)
    idx_path = f"/tmp/svdss_bench_{ref_total}.fmd"
    if rank == 0:
        ix = svdss_amd.FMDIndex.build(ref)
        if world > 1:
            ix.save(idx_path)
    if world > 1:
        dist.barrier()
        if rank != 0:
            ix = svdss_amd.FMDIndex.load(idx_path)
    ix.to_device(local_rank)
    t_index = time.time() - t0

    # ---- reads: generated on the GPU, one independent shard per rank -------
    ref_t = torch.from_numpy(ref[0] if len(ref) == 1 else np.concatenate(ref)).to(device)
    del ref
    d_reads, d_offs = simulate_reads_gpu(ref_t, n_reads, L, args.err, seed=13 + 1000 * rank, device=device)
    total_syms = n_reads * L
    del ref_t
    torch.cuda.synchronize()

    pp = svdss_amd.PingPong(ix, assemble=True)
    stream = torch.cuda.current_stream()

    def step():
        pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), n_reads, total_syms,
                                   stream=stream.cuda_stream, fetch=False)
        if world > 1 and args.gather:
            counts, qs, ln = pp.device_results()
            multi.gather_sfs(counts, qs, ln)

    # raw (unassembled) SFS count: the N_sfs of the algorithmic-bytes formula
    pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), n_reads, total_syms,
                               stream=stream.cuda_stream, assemble=False, fetch=False)
    n_sfs_raw = pp.last_total
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, pipeline_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(pp.last_search_kernel_ms)   # HIP events recorded on `stream` around the search kernel alone
        pipeline_ms.append(pp.last_kernel_ms)        # ... and around order + search + stitch + assemble
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ext = pp.last_total_ext
    n_sfs_asm = pp.last_total
    # algorithmic bytes per read = N_ext*64 + L + 16*N_sfs
    alg_bytes = n_ext * 64 + total_syms + 16 * n_sfs_raw
    k_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "reads/sec through SFS search (ping-pong FMD search + assemble), HiFi 15 kb reads",
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
                "workload": (f"synthetic {ref_total} bp reference in {len(contig_lens)} contig(s) "
                             f"({'GRCh38 primary lengths' if len(contig_lens) == 24 else 'chr20 length' if ref_total == 64_444_167 else 'custom'}"
                             f", iid ACGT + 3% diverged repeats, both strands indexed: {ix.size} BWT symbols), "
                             f"{n_reads} reads/GPU/step x {L} bp ({n_reads * L / ref_total:.2f}x per step), "
                             f"{args.err * 100:.2f}% errors, search with fused assemble, all reads searched "
                             "(--noputative semantics)"),
                "reads_per_gpu": n_reads, "read_len": L, "index_bytes": ix.device_bytes,
                "parallelism": (f"reads sharded over {world} GPU(s), index replicated, no data-path collective"
                                + (", SFS gathered on rank 0 every step" if (world > 1 and args.gather) else "")),
                "ext_per_read": n_ext / n_reads, "raw_sfs_per_read": n_sfs_raw / n_reads,
                "assembled_sfs_per_read": n_sfs_asm / n_reads, "index_build_s": round(t_index, 1),
                "kmer_table_k": ix.kmer_k, "segments_per_read": pp.last_segments,
                "reads_redone_unsegmented": pp.last_fallbacks,
            },
            "roofline": {
                "bound": "hbm", "kernel": "sfs_search2_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(ref_total, n_reads, L, ix.kmer_k),
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms,
                # measured HBM-side rate of the same kernel (traffic / kernel time) next to the spec peak
                "traffic_gbs": (measured_traffic(ref_total, n_reads, L, ix.kmer_k) or 0) / (k_ms * 1e-3) / 1e9 or None,
                "traffic_frac_of_peak": ((measured_traffic(ref_total, n_reads, L, ix.kmer_k) or 0) / (k_ms * 1e-3) / 1e9
                                         / HBM_PEAK_GBS) or None,
                "note": ("achieved = SURVEY 8(d) algorithmic bytes (one 64-B BWT block per rb3_fmd_extend the reference "
                         "would make) / search-kernel time; the k-mer table, text-compare and SET operations answer "
                         "most of those extensions without fetching their blocks, so frac exceeds 1 -- `traffic` "
                         "(TCC_EA0_RDREQ_128B x 128 B + writes: every random 16-B read moves a 128-B line) is what really "
                         "crosses the fabric, `traffic_frac_of_peak` its share of the 8 TB/s peak, and `random_access` "
                         "the limit that binds"),
                "all_kernels_ms": float(np.mean(pipeline_ms)),
                # the kernel's memory operations are dependent random reads (one per lane per iteration): the
                # measured ceiling of that access pattern on this GPU (tools/random_read_probe.hip) next to the
                # rate at which the kernel's measured traffic arrives, both in memory requests (128-B lines) per second
                "random_access": random_access_info(measured_traffic(ref_total, n_reads, L, ix.kmer_k, "_transactions"),
                                                    measured_traffic(ref_total, n_reads, L, ix.kmer_k, "_read_transactions"), k_ms),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ix, d_reads, L, n_reads, args.cpu_seconds)
        if world == 1 and not args.no_call_dp:
            out["config"]["call_dp"] = call_dp_throughput(local_rank)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def random_access_info(transactions, read_transactions, k_ms):
    try:
        with open(os.path.join(ROOT, "profiles", "random_access.json")) as fh:
            probe = json.load(fh)
    except OSError:
        return None
    out = {"ceiling_dependent_lines_per_s": probe["dependent_random_lines_per_s"],
           "ceiling_independent_lines_per_s": probe["independent_random_lines_per_s"], "source": probe["source"]}
    if transactions:   # TCC_EA0_RDREQ + TCC_EA0_WRREQ of one launch (profiles/r01_final_pmc.csv)
        rate = transactions / (k_ms * 1e-3)
        out["achieved_transactions_per_s"] = rate
    if read_transactions:   # the probe measures reads: compare reads with reads
        rrate = read_transactions / (k_ms * 1e-3)
        out["achieved_read_transactions_per_s"] = rrate
        out["read_frac_of_dependent_ceiling"] = rrate / probe["dependent_random_lines_per_s"]
        out["read_frac_of_independent_ceiling"] = rrate / probe["independent_random_lines_per_s"]
    return out


def measured_traffic(ref_total, n_reads, L, k, suffix=""):
    """HBM-side bytes per launch of the search kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, profiles/traffic.json), for exactly this workload; None if that
    configuration has not been profiled.  Counters cannot be read from inside the timed run."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            table = json.load(fh)
    except OSError:
        return None
    return table.get(f"ref{ref_total}_reads{n_reads}_len{L}_k{k}{suffix}")


def call_dp_throughput(device):
    """Side measurement (not part of `value`): the call-stage DP kernels on synthetic sub-clusters of the
    SURVEY 2.3 shapes -- 1024 sub-clusters x 12 reads x ~1 kb with 1% errors and one 150-bp insertion in
    half of them: POA consensus, consensus->reference realignment (full matrix + traceback), and the
    chain filter's ratio on the resulting alleles.  Integer DP: reported in cell updates per second."""
    from svdss_amd import caller
    rng = np.random.default_rng(99)
    clusters, refs = [], []
    for c in range(1024):
        ln = int(rng.integers(600, 1400))
        t = rng.integers(0, 4, size=ln).astype(np.uint8)
        alt = np.concatenate([t[:ln // 2], rng.integers(0, 4, size=150).astype(np.uint8), t[ln // 2:]]) if c % 2 else t
        reads = []
        for _ in range(12):
            r = alt.copy()
            e = rng.random(len(r)) < 0.01
            r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
            reads.append(r)
        clusters.append(reads)
        refs.append(t)
    cons, poa = caller.run_poa(clusters, device=device)
    scores, cigars, aln = caller.ksw_extd2_global(cons, refs, device=device)
    t0 = time.perf_counter()
    ratio, _ = caller.fuzz_ratio(cons[:-1], cons[1:], device=device)
    t_ratio = time.perf_counter() - t0
    n_ins = sum(1 for cg in cigars if any((int(x) & 0xf) == 1 and (int(x) >> 4) >= 100 for x in cg))
    return {"subclusters": len(clusters), "reads_per_subcluster": 12,
            "poa_cells": poa["cells"], "poa_kernel_ms": round(poa["kernel_ms"], 3),
            "poa_gcups": poa["cells"] / (poa["kernel_ms"] * 1e-3) / 1e9,
            "realign_cells": aln["cells"], "realign_kernel_ms": round(aln["kernel_ms"], 3),
            "realign_gcups": aln["cells"] / (aln["kernel_ms"] * 1e-3) / 1e9,
            "ratio_pairs": len(ratio), "ratio_wall_ms": round(t_ratio * 1e3, 3),
            "insertions_recovered": n_ins}


def cpu_baseline(ix, d_reads, L, n_reads, target_s):
    """The oracle (CPU restatement of ping_pong.cpp:4-49 + assembler.cpp:34-56, OpenMP over reads
    like ping_pong.cpp:329) timed on this box's cores on a bounded sample of the same reads."""
    from tests import oracle_lib as O
    fm = O.OracleFMD.from_bwt(ix.bwt())
    threads = O.max_threads()

    def run(k):
        flat = d_reads[:k * L].cpu().numpy()
        offs = np.arange(k + 1, dtype=np.int64) * L
        t0 = time.perf_counter()
        fm.search_batch(flat, offs, True, threads)
        return time.perf_counter() - t0

    k = min(n_reads, 4 * threads)
    t = run(k)
    k2 = int(min(n_reads, max(k, k * target_s / max(t, 1e-3))))
    t2 = run(k2)
    return {"value": k2 / t2, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": f"first {k2} reads of the rank-0 batch, {t2:.1f} s, oracle/svdss_oracle.c "
                      f"orc_search_batch with {threads} OpenMP threads (plain sampled-Occ FMD, faster than "
                      "ropebwt3's rld0: the GPU/CPU ratio is conservative)"}


if __name__ == "__main__":
    main()
