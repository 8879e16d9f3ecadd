"""Host-side mirror of `SVDSS call` stages 1-3 (/root/reference/clusterer.cpp): place every SFS on
the reference through its read's CIGAR, extend to unique flanking k-mers, cluster by proximity,
collect per-cluster coverage and read sub-sequences.  Integer/interval bookkeeping only (SURVEY
8(a) rows a10-a12); thread-count-dependent orderings of the reference are replayed with an
explicit `threads` parameter (SURVEY App. A#7,#8).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from .caller import Cluster, SubRead

BAM_CMATCH, BAM_CINS, BAM_CDEL, BAM_CREF_SKIP, BAM_CSOFT_CLIP, BAM_CHARD_CLIP, BAM_CPAD, BAM_CEQUAL, BAM_CDIFF = range(9)
FLAG_UNMAP, FLAG_SECONDARY, FLAG_SUPPLEMENTARY = 4, 256, 2048


@dataclass
class Alignment:
    """The bam1_t fields `call` consumes."""
    qname: str
    flag: int
    tid: int
    pos: int
    mapq: int
    cigar: List[Tuple[int, int]]      # (length, op code) like decode_cigar (bam.cpp:25-35)
    seq: str
    tags: Dict[str, int] = field(default_factory=dict)
    qual: bytes = b""

    def endpos(self) -> int:          # bam_endpos
        ref = sum(l for l, op in self.cigar if op in (BAM_CMATCH, BAM_CDEL, BAM_CREF_SKIP, BAM_CEQUAL, BAM_CDIFF))
        return self.pos + (ref if ref else 1)


@dataclass
class ExtSFS:
    """SFS after placement (7-argument constructor, sfs.hpp:52-62)."""
    chrom: str
    qname: str
    rs: int
    re: int
    qs: int
    qe: int
    htag: int

    @property
    def l(self):
        return self.qe - self.qs + 1


def get_aligned_pairs(aln: Alignment) -> List[Tuple[int, int]]:
    """bam.cpp:92-134: pysam-like (qpos, rpos) list, -1 on the gapped side."""
    res = []
    ref_pos, read_pos = aln.pos, 0
    for l, op in aln.cigar:
        if op in (BAM_CMATCH, BAM_CEQUAL, BAM_CDIFF):
            for i in range(l):
                res.append((read_pos + i, ref_pos + i))
            read_pos += l
            ref_pos += l
        elif op in (BAM_CINS, BAM_CSOFT_CLIP):
            for i in range(l):
                res.append((read_pos + i, -1))
            read_pos += l
        elif op in (BAM_CDEL, BAM_CREF_SKIP):
            for i in range(l):
                res.append((-1, ref_pos + i))
            ref_pos += l
        # hard clip / pad: advance neither
    return res


def get_unique_kmers(alpairs, k: int, from_end: bool, chrom_seq: str) -> Tuple[int, int]:
    """clusterer.cpp:351-405 (fall-through behaviour of SURVEY App. A#20 included)."""
    n = len(alpairs)
    if n < k:
        return (-1, -1)
    kmers: Dict[str, int] = {}
    i = 0
    while i < n - k + 1:
        skip = False
        for j in range(i, i + k):
            if alpairs[j][0] == -1 or alpairs[j][1] == -1:
                skip = True
                i = j + 1
                break
        if skip:
            continue
        r = alpairs[i][1]
        kmer = chrom_seq[r:r + k]
        kmers[kmer] = kmers.get(kmer, 0) + 1
        i += 1
    last = (-1, -1)
    i = 0
    while i < n - k + 1:
        offset = n - k - i if from_end else i
        skip = False
        for j in range(offset, offset + k):
            if alpairs[j][0] == -1 or alpairs[j][1] == -1:
                skip = True
                i += j - offset
                break
        if skip:
            i += 1
            continue
        last = alpairs[offset]
        r = alpairs[offset][1]
        if kmers.get(chrom_seq[r:r + k], 0) == 1:
            break
        i += 1
    return last


class Clusterer:
    def __init__(self, sfs_by_read: Dict[str, List[Tuple[int, int, int]]], chromosomes: Dict[str, str],
                 ref_names: Sequence[str], threads: int = 4, min_mapq: int = 20, min_cluster_weight: int = 2,
                 flank: int = 100, ksize: int = 7, bsize: int = 10000):
        self.SFSs = sfs_by_read            # qname -> [(qs, l, htag)] in .sfs file order (sfs.cpp:5-30)
        self.chromosomes = chromosomes
        self.ref_names = list(ref_names)
        self.threads, self.min_mapq, self.min_cluster_weight = threads, min_mapq, min_cluster_weight
        self.flank, self.ksize = flank, ksize
        self.bsize = (bsize // threads) * threads   # config.cpp:106
        self.unplaced = self.s_unplaced = self.e_unplaced = self.unknown = 0
        self.unextended = self.small_clusters = self.small_clusters_2 = 0
        self.extended_SFSs: List[ExtSFS] = []
        self.clusters: List[Cluster] = []

    # ---- a10 ------------------------------------------------------------
    def extend_alignment(self, aln: Alignment) -> List[ExtSFS]:
        """clusterer.cpp:159-346 for one read (soft-clip bookkeeping of --clipped omitted)."""
        chrom = self.ref_names[aln.tid]
        if chrom not in self.chromosomes:
            return []
        cseq = self.chromosomes[chrom]
        alpairs = get_aligned_pairs(aln)
        k = self.ksize
        last_pos = 0
        local: List[ExtSFS] = []
        for (qs0, l0, htag) in self.SFSs[aln.qname]:
            s, e = qs0, qs0 + l0 - 1
            aln_start = aln_end = -1
            refs = refe = -1
            for i in range(last_pos, len(alpairs)):
                q, r = alpairs[i]
                if q == -1 or r == -1:
                    continue
                elif q < s:
                    last_pos = i
                    refs = r
                    aln_start = i
                elif q > e:
                    refe = r
                    aln_end = i
                    break
            if refs == -1 and refe == -1:
                self.unplaced += 1
                continue
            elif refs == -1:
                self.s_unplaced += 1
                continue
            elif refe == -1:
                self.e_unplaced += 1
                continue
            local_alpairs = []
            last_r = refs - 1
            for i in range(aln_start, aln_end + 1):
                q, r = alpairs[i]
                if r == -1:
                    if refs <= last_r <= refe:
                        local_alpairs.append((q, r))
                else:
                    last_r = r
                    if refs <= r <= refe:
                        local_alpairs.append((q, r))
                if q != -1 and r != -1 and r >= refe:
                    break
            pre = []
            for i in range(aln_start - 1, -1, -1):
                pre.append(alpairs[i])
                if len(pre) == self.flank:
                    break
            pre.reverse()
            post = []
            for i in range(aln_end + 1, len(alpairs)):
                post.append(alpairs[i])
                if len(post) == self.flank:
                    break
            prekmer = get_unique_kmers(pre, k, True, cseq)
            postkmer = get_unique_kmers(post, k, False, cseq)
            if prekmer[0] == -1 or prekmer[1] == -1:
                prekmer = local_alpairs[0]
            if postkmer[0] == -1 or postkmer[1] == -1:
                postkmer = local_alpairs[-1]
            if prekmer[0] == -1 or prekmer[1] == -1 or postkmer[0] == -1 or postkmer[1] == -1:
                self.unknown += 1
                continue
            if prekmer[1] > postkmer[1] + k:
                continue                                  # warning only in the reference (:301-303)
            local.append(ExtSFS(chrom, aln.qname, prekmer[1], postkmer[1] + k, prekmer[0], postkmer[0] + k, htag))
        merged: List[ExtSFS] = []                         # single-pass, first-match merge (:314-336)
        for x in local:
            for m in merged:
                if (x.rs <= m.rs <= x.re) or (m.rs <= x.rs <= m.re):
                    m.rs, m.re = min(m.rs, x.rs), max(m.re, x.re)
                    m.qs, m.qe = min(m.qs, x.qs), max(m.qe, x.qe)
                    break
            else:
                merged.append(ExtSFS(x.chrom, x.qname, x.rs, x.re, x.qs, x.qe, x.htag))
        return merged

    def align_and_extend(self, alignments: Sequence[Alignment]):
        """clusterer.cpp:56-156: batches of bsize eligible reads, read n of a batch on thread n % T;
        extended SFS concatenated thread by thread (:21-25)."""
        T = self.threads
        per_thread: List[List[ExtSFS]] = [[] for _ in range(T)]
        eligible = [a for a in alignments
                    if not (a.flag & (FLAG_UNMAP | FLAG_SUPPLEMENTARY | FLAG_SECONDARY))
                    and a.mapq >= self.min_mapq and a.qname in self.SFSs]
        for b0 in range(0, len(eligible), max(self.bsize, 1)):
            batch = eligible[b0:b0 + self.bsize]
            for t in range(T):
                for a in batch[t::T]:
                    per_thread[t].extend(self.extend_alignment(a))
        self.extended_SFSs = [x for t in range(T) for x in per_thread[t]]

    # ---- a11 ------------------------------------------------------------
    def cluster_by_proximity(self) -> List[List[ExtSFS]]:
        """clusterer.cpp:407-474; returns the SFS groups in the order Clusterer::run pushes them
        (:33-35: thread by thread, std::map<(low,high)> order inside a thread; the key has no chrom)."""
        ext = sorted(self.extended_SFSs, key=lambda s: (s.chrom, s.rs))   # sfs.hpp:66-73 (stable here)
        self.extended_SFSs = ext
        if not ext:
            return []
        dist = int((max(s.re - s.rs for s in ext)) * 1.1)
        intervals = []
        prev_i, prev_e, prev_chrom = 0, ext[0].re, ext[0].chrom
        for i in range(1, len(ext)):
            s = ext[i]
            if s.chrom != prev_chrom:
                prev_chrom = s.chrom
                intervals.append((prev_i, i - 1))
                prev_i, prev_e = i, s.re
                continue
            if s.rs - prev_e > dist:
                intervals.append((prev_i, i - 1))
                prev_e, prev_i = s.re, i
        intervals.append((prev_i, len(ext) - 1))
        T = self.threads
        per_thread: List[Dict[Tuple[int, int], List[ExtSFS]]] = [dict() for _ in range(T)]
        for i, (a, b) in enumerate(intervals):
            t = i % T                                     # schedule(static, 1)
            j = a
            low, high, last_j = ext[j].rs, ext[j].re, j
            j += 1
            while j <= b:
                s = ext[j]
                if s.rs <= high:
                    low, high = min(low, s.rs), max(high, s.re)
                else:
                    per_thread[t].setdefault((low, high), []).extend(ext[last_j:j])
                    low, high, last_j = s.rs, s.re, j
                j += 1
            per_thread[t].setdefault((low, high), []).extend(ext[last_j:b + 1])
        groups = []
        for t in range(T):
            for key in sorted(per_thread[t]):
                groups.append(per_thread[t][key])
        return groups

    # ---- a12 ------------------------------------------------------------
    def fill_clusters(self, groups: List[List[ExtSFS]], alignments: Sequence[Alignment]) -> List[Cluster]:
        """clusterer.cpp:477-610.  The BAI region query `chrom:min_s-max_e` (0-based values in a 1-based
        inclusive region) is replayed as a scan in file order for alignments
        overlapping the 0-based half-open interval [min_s-1, max_e)."""
        by_tid: Dict[int, List[Alignment]] = {}
        for a in alignments:
            by_tid.setdefault(a.tid, []).append(a)
        clusters = []
        for sfss in groups:
            c = Cluster(sfss[0].chrom)
            c.SFSs = sfss
            clusters.append(c)
            reads = {s.qname for s in sfss}
            min_s = min(s.rs for s in sfss)
            max_e = max(max(s.re for s in sfss), 0)
            if len(reads) < self.min_cluster_weight:
                self.small_clusters += 1
                continue
            c.s, c.e = min_s, max_e
            coverages = [0, 0, 0]
            locus_reads = []
            if c.chrom not in self.ref_names:
                continue
            tid = self.ref_names.index(c.chrom)
            beg0, end0 = max(min_s - 1, 0), max_e
            for aln in by_tid.get(tid, []):
                if not (aln.pos < end0 and aln.endpos() > beg0):
                    continue
                if aln.flag & (FLAG_UNMAP | FLAG_SUPPLEMENTARY | FLAG_SECONDARY):
                    continue
                if aln.mapq < self.min_mapq:
                    continue
                hp = aln.tags.get("HP", 0)
                coverages[hp] += 1
                locus_reads.append([0, 3 if hp == 0 else hp])
                if aln.qname not in reads:
                    continue
                locus_reads[-1][0] = 1
                alpairs = get_aligned_pairs(aln)
                qs = qe = -1
                for q, r in reversed(alpairs):
                    if q == -1 or r == -1:
                        continue
                    if r <= min_s:
                        qs = q
                        break
                for q, r in alpairs:
                    if q == -1 or r == -1:
                        continue
                    if r >= max_e:
                        qe = q
                        break
                if qs == -1 or qe == -1:
                    self.unextended += 1
                else:
                    c.add_subread(SubRead(aln.qname, aln.seq[qs:qe + 1], hp))
            if c.size() >= self.min_cluster_weight:
                c.cov0, c.cov1, c.cov2 = coverages
                c.cov = sum(coverages)
                c.reads = [tuple(x) for x in locus_reads]
            else:
                self.small_clusters_2 += 1
        self.clusters = clusters
        return clusters

    def run(self, alignments: Sequence[Alignment]) -> List[Cluster]:
        """Clusterer::run (clusterer.cpp:8-52)."""
        self.align_and_extend(alignments)
        groups = self.cluster_by_proximity()
        return self.fill_clusters(groups, alignments)
