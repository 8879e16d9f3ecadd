"""a13: Caller::split_cluster (/root/reference/caller.cpp:78-255) host logic, hand-worked cases."""
from tests.mirror.caller import Cluster, SubRead, split_cluster, split_cluster_by_len


def _cluster(lens_tags, cov=(12, 4, 5, 3)):
    c = Cluster("chr1", 100, 900, *cov)
    for i, (l, t) in enumerate(lens_tags):
        c.add_subread(SubRead(f"r{i}", "A" * l, t))
    c.reads = [(1, 0)]
    return c


def test_by_len_first_fit_against_integer_mean():
    # 1000 -> bucket0 (mean 1000); 980: 980/1000 = .98 >= .97 joins (mean 990); 955: 955/990 = .9646 < .97 -> bucket1;
    # 1015: 990/1015 = .975 joins bucket0 (mean (1000+980+1015)/3 = 998)
    subs = split_cluster_by_len(_cluster([(1000, 0), (980, 0), (955, 0), (1015, 0)]))
    assert [[sr.size() for sr in s.subreads] for s in subs] == [[1000, 980, 1015], [955]]
    assert subs[0].get_len() == 998 and subs[0].cov == 12 and subs[1].chrom == "chr1"


def test_untagged_cluster_keeps_two_largest_buckets():
    c = _cluster([(500, 0)] * 3 + [(800, 0)] * 4 + [(650, 0)] * 2 + [(300, 0)])
    out = split_cluster(c)
    assert [(o.size(), o.subreads[0].size()) for o in out] == [(4, 800), (3, 500)]
    assert out[0].cov1 == -1 and out[0].cov2 == -1 and out[0].cov0 == 4
    # --noht: tags ignored
    c = _cluster([(500, 1)] * 3 + [(800, 2)] * 4)
    out = split_cluster(c, useht=False)
    assert [o.size() for o in out] == [4, 3]


def test_both_haplotypes_tagged():
    # hap1: 3 x 700; hap2: 2 x 705 and 1 x 400 -> largest bucket per haplotype.  An untagged read that fits a
    # bucket on BOTH sides is compared through int-truncated ratios: r in [.97,1) -> 0,
    # r == 1.0 -> 1, so it joins a side only on an exact length match there; otherwise 0 > 0 fails both ways and
    # the read is dropped.  A read that fits one side only beats the other side's initial -1.
    c = _cluster([(700, 1)] * 3 + [(705, 2), (705, 2), (400, 2)] + [(700, 0), (702, 0), (705, 0), (410, 0), (2000, 0)])
    out = split_cluster(c)
    assert len(out) == 2
    h1, h2 = out
    assert [sr.size() for sr in h1.subreads] == [700, 700, 700, 700]   # exact 700 joined (1 > 0); 702 dropped
    assert h1.cov1 == 5 + 1 and h1.cov0 == -1 and h1.cov2 == -1
    assert [sr.size() for sr in h2.subreads] == [705, 705, 705]         # exact 705 joined
    assert h2.cov2 == 3 + 1
    # 410 fits only the hap2 bucket of 400 (ratio .9756 -> int 0 > -1): it joined that bucket, which is not
    # the largest hap2 bucket and therefore not reported; 2000 fits nothing and is dropped


def test_single_haplotype_plus_rest():
    # only hap1 tagged: untagged reads that fit a hap1 bucket join it, the rest form their own best bucket
    c = _cluster([(600, 1), (610, 1)] + [(605, 0), (300, 0), (305, 0), (1200, 0)])
    out = split_cluster(c)
    assert len(out) == 2
    assert [sr.size() for sr in out[0].subreads] == [600, 610, 605] and out[0].cov1 == 5 + 1
    assert [sr.size() for sr in out[1].subreads] == [300, 305]
    assert out[1].cov0 == 4 - 1 and out[1].cov1 == -1                  # new_cluster.cov0 decremented once
    # copies made inside split_cluster do not carry `reads` (clusterer.hpp:49-59)
    assert out[0].reads == []
