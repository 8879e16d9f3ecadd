"""Python restatement of the HOST logic of `SVDSS call` after the DP seams (test infrastructure: the checker of
svdss_amd/csrc/call_host.cpp; the product is the C++ binary).  Caller::pcall after run_poa, SV extraction, clean_dups,
filter_sv_chains, split_cluster and the VCF writer (/root/reference/caller.cpp:78-475, sv.cpp).  The DP itself
(POA, ksw_extd2, fuzz::ratio) runs on the GPU through svdss_amd.calldp -- re-exported here so that the tests read like
the reference's Caller."""
import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from svdss_amd._lib import check, lib
from svdss_amd.calldp import (CHAR26, GAPE, GAPE2, GAPO, GAPO2, KSW_MAT, SC_MCH, SC_MIS, cigar_string, encode26,  # noqa: F401
                              fuzz_ratio, ksw_extd2_global, run_poa)
from svdss_amd.pingpong import pack_reads  # noqa: F401

# ---------------------------------------------------------------------------
# Host logic of Caller::pcall after run_poa, clean_dups, filter_sv_chains and the VCF writer
# (/root/reference/caller.cpp:326-475, sv.cpp, caller.cpp:477-550).  The DP inside
# (ksw_extd2_sse, fuzz::ratio) runs on the GPU through the functions above; everything else
# is the reference's own integer/string bookkeeping.

class SV:
    """sv.hpp:14-59 / sv.cpp:7-27."""

    def __init__(self, type_, chrom, s, refall, altall, w, cov, ngaps, score, imprecise=False, l=0, cigar="."):
        self.type, self.chrom, self.s = type_, chrom, int(s)
        self.refall, self.altall = refall, altall
        self.e = self.s + len(refall) - 1                       # sv.cpp:15
        self.w, self.l, self.cov = int(w), int(l), int(cov)
        self.cov0 = self.cov1 = self.cov2 = 0
        self.ngaps, self.score, self.imprecise, self.cigar = int(ngaps), int(score), bool(imprecise), cigar
        self.idx = f"{type_}_{chrom}:{self.s}-{self.e}_{abs(self.l)}"   # sv.cpp:23-24
        self.gt, self.gtq = "./.", 0
        self.rvec, self.reads = "", ""

    def add_reads(self, names):                                  # sv.cpp:29-33
        self.reads = ",".join(names)

    def set_cov(self, cov, cov0, cov1, cov2):                    # sv.cpp:35-40
        self.cov, self.cov0, self.cov1, self.cov2 = cov, cov0, cov1, cov2

    def set_rvec(self, reads):                                   # sv.cpp:42-46
        self.rvec = "-".join(f"{a}:{b}" for a, b in reads)

    def set_gt(self, gt, gtq):                                   # sv.cpp:48-51
        self.gt, self.gtq = gt, gtq

    def key(self):                                               # sv.hpp:47-55 operator<
        return (self.chrom, self.s)

    def vcf_line(self) -> str:                                   # sv.cpp:53-80
        svlen = -self.l if self.type == "DEL" else self.l
        return (f"{self.chrom}\t{self.s}\t{self.idx}\t{self.refall}\t{self.altall}\t.\tPASS\t"
                f"VARTYPE=SV;SVTYPE={self.type};SVLEN={svlen};END={self.e};WEIGHT={self.w};COV={self.cov};"
                f"COV0={self.cov0};COV1={self.cov1};COV2={self.cov2};AS={self.score};NV={self.ngaps};"
                f"CIGAR={self.cigar};RVEC={self.rvec};READS={self.reads}"
                + (";IMPRECISE\t" if self.imprecise else "\t") + f"GT:GQ\t{self.gt}:{self.gtq}")


def extract_svs(chrom, chrom_seq: str, cl_s: int, consensus: str, cigar_ops, score: int, min_sv_length: int,
                cl_size: int, cov, names, rvec_reads):
    """caller.cpp:357-401: walk the consensus->reference CIGAR, one SV per I/D >= min_sv_length.
    cov = (cov, cov0, cov1, cov2) of the sub-cluster; rvec_reads = parent cluster's (has_sfs, hp) list."""
    cig = cigar_string(cigar_ops)
    rpos, cpos, nv = int(cl_s), 0, 0
    out = []
    for c in cigar_ops:
        l, op = int(c) >> 4, "MID"[int(c) & 0xf]
        if op == "M":
            rpos += l
            cpos += l
        elif op == "I":
            if l >= min_sv_length:
                anchor = chrom_seq[rpos - 1]
                sv = SV("INS", chrom, rpos, anchor, anchor + consensus[cpos:cpos + l], cl_size, cov[0], nv, score,
                        False, l, cig)
                sv.add_reads(names)
                out.append(sv)
                nv += 1
            cpos += l
        else:
            if l >= min_sv_length:
                sv = SV("DEL", chrom, rpos, chrom_seq[rpos - 1:rpos + l], chrom_seq[rpos - 1], cl_size, cov[0], nv,
                        score, False, l, cig)
                sv.add_reads(names)
                out.append(sv)
                nv += 1
            rpos += l
    for sv in out:
        sv.ngaps = nv
        sv.set_gt("0/1", 100)
        sv.set_cov(*cov)
        sv.set_rvec(rvec_reads)
    return out


def consensus_sam_row(chrom: str, s: int, e: int, cigar: str, seq: str) -> str:
    """operator<< of Consensus (caller.hpp:56-69): one SAM row of the --poa output."""
    return f"{chrom}:{s + 1}-{e + 1}\t0\t{chrom}\t{s + 1}\t60\t{cigar}\t*\t0\t0\t{seq}\t*"


def sam_text(contigs, rows) -> str:
    """Caller::write_sam (caller.cpp:65-75)."""
    return "@HD\tVN:1.4\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in contigs) + "".join(r + "\n" for r in rows)


def pcall_tail(subclusters, chromosomes: dict, min_sv_length: int = 25, threads: int = 4, device: int = 0,
               sam_rows: list = None, alignments=None):
    """Caller::pcall from the consensus on (caller.cpp:326-405) for a list of sub-clusters, each a
    dict(chrom, s, e, consensus, size, names, cov=(cov,cov0,cov1,cov2), rvec, cluster_index).
    All realignments go to the GPU in one batch.  Returns SVs in the order Caller::run leaves them
    before its sort (caller.cpp:18-22: per-thread vectors, cluster i on thread i % T, each inserted
    at the FRONT).  alignments = (scores, cigars) skips the realignment batch.  sam_rows, if given, receives the --poa rows (caller.cpp:356-357) in that same order."""
    if alignments is not None:     # (scores, cigars) already computed, e.g. by the ranks of a sharded call
        scores, cigars = alignments
        scores, stats = np.asarray(scores), {}
    else:
        refs = [chromosomes[sc["chrom"]][sc["s"]:sc["e"] + 1] for sc in subclusters]   # caller.cpp:329
        cons = [sc["consensus"] for sc in subclusters]
        scores, cigars, stats = ksw_extd2_global(cons, refs, device=device)
    per_thread = [[] for _ in range(threads)]
    per_thread_sam = [[] for _ in range(threads)]
    for sc, score, cg in zip(subclusters, scores.tolist(), cigars):
        per_thread_sam[sc.get("cluster_index", 0) % threads].append(
            consensus_sam_row(sc["chrom"], sc["s"], sc["e"], cigar_string(cg), sc["consensus"]))
        svs = extract_svs(sc["chrom"], chromosomes[sc["chrom"]], sc["s"], sc["consensus"], cg, score,
                          min_sv_length, sc["size"], sc["cov"], sc["names"], sc["rvec"])
        per_thread[sc.get("cluster_index", 0) % threads].extend(svs)
    out = []
    for t in range(threads):
        out = per_thread[t] + out
    if sam_rows is not None:
        for t in reversed(range(threads)):
            sam_rows.extend(per_thread_sam[t])
    return out, stats


def clean_dups(svs):
    """caller.cpp:409-426: drop an SV equal (chrom, s, refall, altall) to the one right before it."""
    out, last = [], ("", -1, "", "")
    for sv in svs:
        cur = (sv.chrom, sv.s, sv.refall, sv.altall)
        if cur != last:
            out.append(sv)
        last = cur
    return out


def filter_sv_chains(svs, min_ratio: float = 0.97, device: int = 0, ratio_fn=None):
    """caller.cpp:429-475.  `prev` is always the element right before `sv`, so every
    rapidfuzz::fuzz::ratio call is on an adjacent pair: all candidate pairs are scored in ONE
    GPU batch, then the sequential keep/merge logic replays on those ratios."""
    if len(svs) < 2:
        return list(svs)
    min_ratio = float(np.float32(min_ratio))     # config.hpp:91: float, compared with a double
    cand = []
    for i in range(1, len(svs)):
        prev, sv = svs[i - 1], svs[i]
        if sv.chrom == prev.chrom and sv.s - prev.e < 2 * sv.l and prev.type == sv.type:
            w_r = min(float(sv.w), float(prev.w)) / max(float(sv.w), float(prev.w))
            l_r = min(float(sv.l), float(prev.l)) / max(float(sv.l), float(prev.l))
            if sv.s - prev.s < 100 and w_r >= 0.9 and l_r >= min_ratio:
                cand.append(i)
    sims = {}
    if cand:
        a = [(svs[i].refall if svs[i].type == "DEL" else svs[i].altall) for i in cand]
        b = [(svs[i - 1].refall if svs[i].type == "DEL" else svs[i - 1].altall) for i in cand]
        ratio, _ = (ratio_fn or fuzz_ratio)(a, b, device=device)
        sims = dict(zip(cand, ratio.tolist()))
    out, prev, reset = [], svs[0], False
    for i in range(1, len(svs)):
        if reset:
            reset = False
            prev = svs[i]
            continue
        sv = svs[i]
        if i in sims and sims[i] > 70:
            out.append(sv if sv.w > prev.w else prev)
            reset = True
            continue
        out.append(prev)
        prev = sv
    out.append(prev)          # caller.cpp:472 (also when a reset is pending: end-of-list artefact)
    return out


def vcf_header(contigs) -> str:
    """caller.cpp:477-550; contigs = [(name, length), ...] in FASTA order."""
    lines = ["##fileformat=VCFv4.2",
             "##reference=ftp://ftp.1000genomes.ebi.ac.uk/vol1/ftp/data_collections/HGSVC2/technical/reference/"
             "20200513_hg38_NoALT/hg38.no_alt.fa.gz"]
    lines += [f"##contig=<ID={n},length={l}>" for n, l in contigs]
    lines += ['##FILTER=<ID=PASS,Description="All filters passed">']
    info = [("VARTYPE", "A", "String", "Variant class"), ("SVTYPE", "1", "String", "Variant type"),
            ("SVLEN", "1", "Integer", "Difference in length between REF and ALT alleles"),
            ("END", "1", "Integer", "End position of the variant described in this record"),
            ("WEIGHT", "1", "Integer", "Number of alignments supporting this record"),
            ("COV", "1", "Integer", "Total number of alignments covering this locus"),
            ("COV0", "1", "Integer", "Total number of alignments covering this locus (no HP)"),
            ("COV1", "1", "Integer", "Total number of alignments covering this locus (HP=1)"),
            ("COV2", "1", "Integer", "Total number of alignments covering this locus (HP=2)"),
            ("AS", "1", "Integer", "Alignment score"), ("NV", "1", "Integer", "Number of variations on same consensus"),
            ("IMPRECISE", "0", "Flag", "Imprecise structural variation"),
            ("CIGAR", "A", "String", "CIGAR of consensus"),
            ("READS", ".", "String", "Reads identifiers supporting the call"),
            ("RVEC", ".", "String", "Reads vector used by genotyper")]
    lines += [f'##INFO=<ID={i},Number={n},Type={t},Description="{d}">' for i, n, t, d in info]
    lines += ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype quality">',
              "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDEFAULT"]
    return "\n".join(lines) + "\n"


def call_tail(subclusters, chromosomes: dict, contigs, min_sv_length: int = 25, threads: int = 4,
              min_ratio: float = 0.97, device: int = 0, sam_rows: list = None, alignments=None, ratio_fn=None) -> str:
    """Caller::run from pcall's realignment to the VCF text (caller.cpp:17-29): realign, extract,
    sort, clean_dups, filter_sv_chains, sort, write.  std::sort's order among SVs with equal
    (chrom, POS) is unspecified in the reference; a stable sort is used here."""
    svs, _ = pcall_tail(subclusters, chromosomes, min_sv_length, threads, device, sam_rows, alignments)
    svs.sort(key=SV.key)
    svs = clean_dups(svs)
    svs = filter_sv_chains(svs, min_ratio, device, ratio_fn)
    svs.sort(key=SV.key)
    return vcf_header(contigs) + "".join(sv.vcf_line() + "\n" for sv in svs)


# ---------------------------------------------------------------------------
# a13: Caller::split_cluster_by_len / split_cluster (caller.cpp:78-255) on the structs of
# clusterer.hpp:24-139.  float = IEEE binary32 as in the reference (np.float32 arithmetic).

class SubRead:                                       # clusterer.hpp:24-36
    def __init__(self, name, seq, htag):
        self.name, self.seq, self.htag = name, seq, int(htag)

    def size(self):
        return len(self.seq)


class Cluster:                                       # clusterer.hpp:38-139
    def __init__(self, chrom="", s=0, e=0, cov=0, cov0=0, cov1=0, cov2=0):
        self.chrom, self.s, self.e = chrom, int(s), int(e)
        self.cov, self.cov0, self.cov1, self.cov2 = cov, cov0, cov1, cov2
        self.subreads, self.reads, self.SFSs = [], [], []

    def copy_cleared(self):
        """Cluster c = cluster; c.clear(): the copy constructor drops `reads` (clusterer.hpp:49-59)."""
        c = Cluster(self.chrom, self.s, self.e, self.cov, self.cov0, self.cov1, self.cov2)
        return c

    def add_subread(self, sr):
        self.subreads.append(sr)

    def size(self):
        return len(self.subreads)

    def get_len(self):                               # integer mean (clusterer.hpp:103-111)
        return sum(sr.size() for sr in self.subreads) // len(self.subreads)

    def get_names(self):
        return [sr.name for sr in self.subreads]

    def get_seqs(self):
        return [sr.seq for sr in self.subreads]


_F = np.float32


def _len_ratio(cl, sl):
    cl, sl = _F(cl), _F(sl)
    return min(cl, sl) / max(cl, sl)


def split_cluster_by_len(cluster: Cluster, min_ratio=0.97):
    """caller.cpp:78-97: greedy first fit by length ratio against the bucket's integer mean length."""
    mr = _F(min_ratio)
    subs = []
    for sr in cluster.subreads:
        for sc in subs:
            if _len_ratio(sc.get_len(), sr.size()) >= mr:
                break
        else:
            sc = Cluster(cluster.chrom, cluster.s, cluster.e, cluster.cov, cluster.cov0, cluster.cov1, cluster.cov2)
            subs.append(sc)
        sc.add_subread(sr)
    return subs


def _largest(subs):
    """first bucket of maximal size (strict > scan, caller.cpp:222-229)."""
    v_max, i_max = 0, -1
    for i, sc in enumerate(subs):
        if sc.size() > v_max:
            v_max, i_max = sc.size(), i
    return i_max


def split_cluster(cluster: Cluster, useht=True, min_ratio=0.97):
    """caller.cpp:100-255, including the int-typed best_ratio quirk."""
    mr = _F(min_ratio)
    c0, c1, c2 = cluster.copy_cleared(), cluster.copy_cleared(), cluster.copy_cleared()
    for sr in cluster.subreads:
        if useht and sr.htag == 1:
            c1.add_subread(sr)
        elif useht and sr.htag == 2:
            c2.add_subread(sr)
        else:
            c0.add_subread(sr)
    c0.cov1 = c0.cov2 = -1
    c1.cov0 = c1.cov2 = -1
    c2.cov0 = c2.cov1 = -1
    out = []
    if c1.size() == 0 and c2.size() == 0:
        subs = split_cluster_by_len(c0, min_ratio)
        i1 = i2 = -1
        v1 = v2 = 0
        for i, sc in enumerate(subs):               # top-2 by size (caller.cpp:134-145)
            if sc.size() > v1:
                v2, i2 = v1, i1
                v1, i1 = sc.size(), i
            elif sc.size() > v2:
                v2, i2 = sc.size(), i
        if i1 != -1:
            out.append(subs[i1])
        if i2 != -1:
            out.append(subs[i2])
        return out
    both = (1 if c1.size() > 0 else 0) + (2 if c2.size() > 0 else 0)
    subs1 = split_cluster_by_len(c1, min_ratio)
    subs2 = split_cluster_by_len(c2, min_ratio)
    new_cluster = Cluster(cluster.chrom, cluster.s, cluster.e, cluster.cov, cluster.cov0, -1, -1)

    def best_of(subs, sl):
        best, best_ratio = -1, 0 - 1                 # `int best_ratio = -1`
        for i, sc in enumerate(subs):
            r = _len_ratio(sc.get_len(), sl)
            if r >= mr and r > _F(best_ratio):
                best, best_ratio = i, int(r)         # truncation: 0 unless r == 1.0
        return best, best_ratio

    for sr in c0.subreads:
        b1, r1 = best_of(subs1, sr.size())
        b2, r2 = best_of(subs2, sr.size())
        if both == 1:
            if b1 == -1:
                new_cluster.add_subread(sr)
            else:
                subs1[b1].add_subread(sr)
                subs1[b1].cov1 += 1
                new_cluster.cov0 -= 1
        elif both == 2:
            if b2 == -1:
                new_cluster.add_subread(sr)
            else:
                subs2[b2].add_subread(sr)
                subs2[b2].cov2 += 1
                new_cluster.cov0 -= 1
        else:
            if b1 != -1 and r1 > r2:
                subs1[b1].add_subread(sr)
                subs1[b1].cov1 += 1
                new_cluster.cov0 -= 1
            elif b2 != -1 and r2 > r1:
                subs2[b2].add_subread(sr)
                subs2[b2].cov2 += 1
                new_cluster.cov0 -= 1
            # else: the untagged sub-read is dropped (caller.cpp:211-212)
    i = _largest(subs1)
    if i != -1:
        out.append(subs1[i])
    i = _largest(subs2)
    if i != -1:
        out.append(subs2[i])
    if both != 3:
        news = split_cluster_by_len(new_cluster, min_ratio) if new_cluster.size() else []
        i = _largest(news)
        if i != -1:
            if both == 1:
                news[i].cov1 = -1
            else:
                news[i].cov2 = -1
            out.append(news[i])
    return out


def call(alignments, sfs_text: str, chromosomes: dict, contigs, ref_names, threads: int = 4,
         min_cluster_weight: int = 2, min_sv_length: int = 25, min_mapq: int = 20, useht: bool = True,
         min_ratio: float = 0.97, device: int = 0, shard=None, poa_fn=None, align_fn=None, ratio_fn=None):
    """Caller::run (caller.cpp:3-57) without --clipped: SFS file + alignments + reference -> VCF text; the
    --poa (SAM) and --clusters side outputs are returned as info["sam"] / info["clusters_text"].  Host bookkeeping as in the reference; POA, realignment and the chain
    filter's ratio run on the GPU in three batched calls.  Returns (vcf_text, info dict).

    shard = (rank, world, all_gather) splits the DP work over the ranks of one node (POA / realignment
    batches shard by sub-cluster index with no exchange; sort -> clean_dups -> filter_sv_chains -> sort runs once on the
    gathered rows): every rank runs the host bookkeeping on the whole input (it is deterministic), takes the
    sub-clusters k with k % world == rank through POA and realignment on its GPU, all_gather(list) returns the list of
    every rank's (k, consensus, score, cigar) rows, and every rank finishes the call on the merged rows -- the VCF is
    byte-identical to the single-GPU one.  svdss_amd.multi.call_sharded wraps this for torch.distributed.
    poa_fn / align_fn / ratio_fn replace run_poa / ksw_extd2_global / fuzz_ratio (tests of the sharding on CPU)."""
    from .clusterer import Clusterer
    from svdss_amd.pingpong import parse_sfsfile
    sfs = parse_sfsfile(sfs_text)
    min_sv_length = max(25, min_sv_length)                      # config.cpp:87
    C_ = Clusterer(sfs, chromosomes, ref_names, threads=threads, min_mapq=min_mapq,
                   min_cluster_weight=min_cluster_weight)
    clusters = C_.run(alignments)
    subs = []
    for i, cluster in enumerate(clusters):
        if cluster.size() < min_cluster_weight:                 # caller.cpp:316-317
            continue
        for cl in split_cluster(cluster, useht, min_ratio):
            subs.append((i, cl, cluster))
    rank, world, all_gather = shard if shard is not None else (0, 1, None)
    mine = list(range(rank, len(subs), world))
    consensus, poa_stats = (poa_fn or run_poa)([subs[k][1].get_seqs() for k in mine], device=device) if mine else ([], {})
    alignments = None
    if world > 1 or align_fn is not None:
        if all_gather is None:
            all_gather = lambda rows: [rows]
        refs = [chromosomes[subs[k][1].chrom][subs[k][1].s:subs[k][1].e + 1] for k in mine]   # caller.cpp:329
        sc_, cg_, _ = (align_fn or ksw_extd2_global)(consensus, refs, device=device) if mine else ([], [], {})
        rows = [(k, c, int(x), [int(w) for w in g]) for k, c, x, g in zip(mine, consensus, sc_, cg_)]
        merged = sorted(r for part in all_gather(rows) for r in part)
        assert [r[0] for r in merged] == list(range(len(subs)))
        consensus = [r[1] for r in merged]
        alignments = ([r[2] for r in merged], [np.asarray(r[3], dtype=np.uint32) for r in merged])
    entries = [dict(chrom=cl.chrom, s=cl.s, e=cl.e, consensus=cons, size=cl.size(), names=cl.get_names(),
                    cov=(cl.cov, cl.cov0, cl.cov1, cl.cov2), rvec=parent.reads, cluster_index=i)
               for (i, cl, parent), cons in zip(subs, consensus)]
    sam_rows = []
    vcf = call_tail(entries, chromosomes, contigs, min_sv_length, threads, min_ratio, device, sam_rows, alignments, ratio_fn)
    # Clusterer::store_clusters (clusterer.cpp:613-626)
    clusters_text = "".join(
        f"{c.chrom}:{c.s + 1}-{c.e + 1}\t{c.size()}" + "".join(f"\t{sr.name}:{sr.seq}" for sr in c.subreads) + "\n"
        for c in clusters)
    info = {"sam": sam_text(contigs, sam_rows), "clusters_text": clusters_text,
            "clusters": len(clusters), "subclusters": len(subs), "extended_sfs": len(C_.extended_SFSs),
            "unplaced": (C_.unplaced, C_.s_unplaced, C_.e_unplaced), "poa": poa_stats}
    return vcf, info
