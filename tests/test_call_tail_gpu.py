"""Caller::pcall tail (consensus -> realign -> SV rows -> dedup -> chain filter -> VCF,
/root/reference/caller.cpp:326-475 + sv.cpp) with the DP on the GPU, against hand-derived
expectations and the oracle DP."""
import numpy as np
import pytest

from svdss_amd import synth
from tests.mirror import caller
from tests.mirror.caller import SV
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _ascii(a):
    return synth.to_ascii(a)


def _mk(chromseq, s, e, consensus, size=5, idx=0, names=("r1", "r2"), cov=(9, 3, 4, 2), rvec=((1, 0), (0, 2))):
    return dict(chrom="chr1", s=s, e=e, consensus=consensus, size=size, names=list(names), cov=cov,
                rvec=list(rvec), cluster_index=idx)


def test_extract_ins_and_del_rows():
    rng = np.random.default_rng(3)
    ref = _ascii(rng.integers(1, 5, size=5000).astype(np.uint8))
    ins = _ascii(rng.integers(1, 5, size=60).astype(np.uint8))
    s, e = 1000, 1999
    win = ref[s:e + 1]
    consensus = win[:300] + ins + win[300:600] + win[680:]       # 60-bp INS at s+300, 80-bp DEL at s+600
    svs, stats = caller.pcall_tail([_mk(ref, s, e, consensus)], {"chr1": ref}, min_sv_length=50)
    assert [v.type for v in svs] == ["INS", "DEL"]
    i, d = svs
    # POS = rpos (0-based next reference base == 1-based anchor), anchor base ref[rpos-1] (caller.cpp:373-388)
    assert i.refall == ref[i.s - 1] and i.altall[0] == i.refall and len(i.altall) == 61 and i.l == 60
    assert d.altall == ref[d.s - 1] and d.refall == ref[d.s - 1:d.s + 80] and d.l == 80
    # left-aligned gaps can slide left of the implanted coordinate, never right
    assert s + 300 - 5 <= i.s <= s + 300 and s + 600 - 5 <= d.s <= s + 600
    assert i.e == i.s and d.e == d.s + 80                       # sv.cpp:15
    assert i.idx == f"INS_chr1:{i.s}-{i.e}_60" and d.idx == f"DEL_chr1:{d.s}-{d.e}_80"
    assert i.ngaps == d.ngaps == 2 and i.gt == "0/1" and i.gtq == 100
    sc, cg = O.ksw_extd2_global(caller.encode26(consensus), caller.encode26(win), caller.KSW_MAT)
    assert i.score == d.score == sc and i.cigar == caller.cigar_string(cg)
    f = i.vcf_line().split("\t")
    assert f[0] == "chr1" and f[1] == str(i.s) and f[3] == i.refall and f[4] == i.altall and f[5:7] == [".", "PASS"]
    assert f[7] == (f"VARTYPE=SV;SVTYPE=INS;SVLEN=60;END={i.e};WEIGHT=5;COV=9;COV0=3;COV1=4;COV2=2;AS={sc};NV=2;"
                    f"CIGAR={i.cigar};RVEC=1:0-0:2;READS=r1,r2")
    assert f[8:] == ["GT:GQ", "0/1:100"]
    assert "SVLEN=-80" in d.vcf_line()
    # below min_sv_length nothing is reported
    svs2, _ = caller.pcall_tail([_mk(ref, s, e, consensus)], {"chr1": ref}, min_sv_length=100)
    assert svs2 == []


def test_thread_order_of_pcall():
    # cluster i runs on thread i % T, per-thread vectors are inserted at the FRONT (caller.cpp:18-22)
    rng = np.random.default_rng(4)
    ref = _ascii(rng.integers(1, 5, size=9000).astype(np.uint8))
    subs = []
    for k in range(5):
        s = 500 + 1500 * k
        win = ref[s:s + 800]
        subs.append(_mk(ref, s, s + 799, win[:400] + win[470:], idx=k))
    svs, _ = caller.pcall_tail(subs, {"chr1": ref}, min_sv_length=50, threads=2)
    starts = [v.s for v in svs]
    order = sorted(range(5), key=lambda k: subs[k]["s"])
    by_cluster = {k: [v for v in svs if subs[k]["s"] <= v.s <= subs[k]["e"]][0].s for k in range(5)}
    assert starts == [by_cluster[1], by_cluster[3], by_cluster[0], by_cluster[2], by_cluster[4]]
    assert len(order) == 5


def test_clean_dups_and_chain_filter():
    rng = np.random.default_rng(5)
    allele = _ascii(rng.integers(1, 5, size=120).astype(np.uint8))
    near = allele[:60] + "A" + allele[60:]                       # ratio well above 70
    far = _ascii(rng.integers(1, 5, size=120).astype(np.uint8))   # random: ratio of ~60 on 4 letters

    def ins(s, alt, w, l=None):
        return SV("INS", "chr1", s, "A", "A" + alt, w, 10, 0, 100, False, l or len(alt), "x")

    a, b, c, d = ins(1000, allele, 10), ins(1000, allele, 10), ins(1040, near, 11), ins(1300, far, 10)
    assert [v.s for v in caller.clean_dups([a, b, c, d])] == [1000, 1040, 1300]   # adjacent exact duplicate
    out = caller.filter_sv_chains([a, c, d])
    assert [v.s for v in out] == [1040, 1300]                     # a/c merged (w ratio 10/11 >= 0.9), heavier kept
    # weight ratio below 0.9 blocks the merge (caller.cpp:447-450)
    c2 = ins(1040, near, 20)
    assert [v.s for v in caller.filter_sv_chains([a, c2, d])] == [1000, 1040, 1300]
    # distance >= 100 blocks it too
    c3 = ins(1100, near, 11)
    assert [v.s for v in caller.filter_sv_chains([a, c3])] == [1000, 1100]
    # dissimilar alleles are both kept
    d2 = ins(1030, far, 10)
    ratio, _ = caller.fuzz_ratio([far], [allele])
    assert ratio[0] <= 70
    assert [v.s for v in caller.filter_sv_chains([a, d2])] == [1000, 1030]
    # after a merge the NEXT element becomes prev without being compared, and at the end of the
    # list a pending reset still pushes prev (caller.cpp:437-441,472): 3 similar SVs -> 2 rows
    e3 = ins(1060, near, 11)
    out = caller.filter_sv_chains([a, c, e3])
    assert [v.s for v in out] == [1040, 1060]
    out = caller.filter_sv_chains([a, c])                         # merge on the last pair: artefact keeps svs[0]
    assert len(out) == 2 and out[0].s == 1040 and out[1].s == 1000
    # DEL compares the REF alleles
    d1 = SV("DEL", "chr1", 500, "G" + allele, "G", 4, 8, 0, 50, False, 120, "x")
    d2 = SV("DEL", "chr1", 520, "G" + near, "G", 4, 8, 0, 50, False, 121, "x")
    assert len(caller.filter_sv_chains([d1, d2, d])) == 2


def test_call_tail_vcf_text():
    rng = np.random.default_rng(6)
    ref = _ascii(rng.integers(1, 5, size=6000).astype(np.uint8))
    win = ref[2000:3000]
    sub = _mk(ref, 2000, 2999, win[:500] + win[560:])
    txt = caller.call_tail([sub, dict(sub, cluster_index=1)], {"chr1": ref}, [("chr1", 6000)], min_sv_length=50)
    lines = txt.splitlines()
    assert lines[0] == "##fileformat=VCFv4.2" and lines[2] == "##contig=<ID=chr1,length=6000>"
    assert lines[-2].startswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDEFAULT")
    body = [l for l in lines if not l.startswith("#")]
    assert len(body) == 1 and "SVTYPE=DEL;SVLEN=-60" in body[0]   # the duplicate call was removed
