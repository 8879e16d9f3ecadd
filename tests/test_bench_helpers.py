"""bench.py's host-side helpers (no GPU): the LPT partition of config 5, the gate that ties profiles/traffic.json to the
kernel version it was measured on, the call-side workload of a step (sizes of config 4), and the CPU baseline's batch
oracle against the per-item oracle calls."""
import json
import os

import numpy as np

import bench
from tests import oracle_lib as O
from tests.common import ROOT


def test_lpt_partition_balances_the_grch38_contigs():
    lens = list(bench.GRCH38_PRIMARY)
    for world in (1, 2, 4, 8):
        owner = bench.lpt_partition(lens, world)
        load = [sum(l for l, o in zip(lens, owner) if o == r) for r in range(world)]
        assert sum(load) == sum(lens) and min(load) > 0
        assert max(load) / (sum(lens) / world) < 1.08       # 24 contigs over 8 ranks: within 8 % of even


def test_traffic_is_only_quoted_for_the_kernel_it_was_measured_on(monkeypatch, tmp_path):
    h = bench.search_kernel_hash()
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    key = "ref3088269832_reads1048576_len15000_k16"
    assert table[key]["kernel_hash"] == h, "profiles/traffic.json is stale: re-run the --pmc pass (tools/collect_profiles_r03.sh)"
    assert bench.profiled(3088269832, 1048576, 15000, 16)["read_requests"] > 2e9
    assert bench.profiled(3088269832, 1048576, 15000, 16, "_families0.05") is None      # another workload
    assert bench.profiled(3088269832, 1048576, 15000, 15) is None
    r = bench.search_roofline(3088269832, 1048576, 15000, 16, 100.0, 4e10, 1.5e10, 7e8, 110.0, 67.0)
    assert r["frac"] is not None and abs(r["frac"] - r["traffic"] / 0.1 / 1e9 / 8000.0) < 1e-9 and r["kernel_hash"] == h
    # a kernel source that changed: no fraction is claimed
    monkeypatch.setattr(bench, "search_kernel_hash", lambda: "0" * 16)
    assert bench.profiled(3088269832, 1048576, 15000, 16) is None
    r = bench.search_roofline(3088269832, 1048576, 15000, 16, 100.0, 4e10, 1.5e10, 7e8, 110.0, 67.0)
    assert r["frac"] is None and r["traffic"] is None and "no committed" in r["note"]


def test_call_workload_and_the_batch_oracle():
    n_clusters = max(2, int(round(bench.SVS_30X_WG * (1 << 20) / bench.READS_30X_WG)))
    assert n_clusters == 3395                                  # config 4: 20,000 SVs per 30x
    cw = bench.CallWorkload(8, seed=99)
    assert cw.n_sub == 12 and int(cw.cluster_off[-1]) == 8 * 30   # every other cluster heterozygous: two halves of 15
    assert cw.seqs.max() <= 3 and cw.mat.shape == (25,)
    cl, sc, nc, ra = O.call_batch(cw.seqs, cw.seq_off, cw.cluster_off, cw.refs, cw.ref_off, cw.mat, 4)
    for j in range(cw.n_sub):
        seqs = [cw.seqs[cw.seq_off[i]:cw.seq_off[i + 1]] for i in range(cw.cluster_off[j], cw.cluster_off[j + 1])]
        c = O.poa_consensus(seqs)
        assert len(c) == cl[j]
        s, cg = O.ksw_extd2_global(c, cw.refs[cw.ref_off[j]:cw.ref_off[j + 1]], cw.mat)
        assert s == sc[j] and len(cg) == nc[j]
        if j:
            assert ra[j - 1] == O.fuzz_ratio(bytes(prev), bytes(c))
        prev = c
    # thread count never changes the result
    assert all((a == b).all() for a, b in zip((cl, sc, nc, ra), O.call_batch(cw.seqs, cw.seq_off, cw.cluster_off, cw.refs, cw.ref_off, cw.mat, 1)))


def test_cpu_quota_is_positive():
    assert 1 <= bench.cpu_quota() <= (os.cpu_count() or 1)
