"""Parity tests proper: the HIP kernels, called through the C-ABI, against the
oracle, the golden vectors and size-independent properties.  Integer work:
every comparison is bit-exact."""
import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from tests import oracle_lib as O
from tests.common import from_ascii, load_golden, small_workload, split

pytestmark = pytest.mark.gpu


def _search(ix, flat, offs, assemble):
    pp = svdss_amd.PingPong(ix, assemble=assemble)
    b = pp.ping_pong_search(flat, offs)
    pp.close()
    return b


def test_library_is_the_hip_build():
    import torch
    assert torch.cuda.is_available()
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_golden_vectors_on_gpu():
    for case in load_golden():
        contigs = [from_ascii(c) for c in case["contigs"]]
        ix = svdss_amd.FMDIndex.build(contigs, threads=2).to_device(0)
        reads = [from_ascii(r["read"]) for r in case["reads"]]
        flat, offs = svdss_amd.pack_reads(reads)
        raw = _search(ix, flat, offs, False)
        for got, rd, ne in zip(raw.per_read(), case["reads"], raw.n_ext.tolist()):
            assert [list(x) for x in got] == rd["sfs"], case["name"]
            assert ne == rd["n_ext"]
        asm = _search(ix, flat, offs, True)
        for got, rd in zip(asm.per_read(), case["reads"]):
            assert [list(x) for x in got] == rd["assembled"], case["name"]


@pytest.mark.parametrize("k,segments", [(0, 1), (0, 4), (12, 1), (12, 4), (16, 1), (16, 4)])
def test_golden_vectors_for_every_table_order_and_launch_shape(k, segments, monkeypatch):
    """The committed vectors (219 brute-force reads: tests/golden/make_golden.py) through the HIP path without a k-mer
    table, with one of order 12 and with the order the whole-genome index uses (16: every case's table is 64 GiB; the
    two largest cases only), one lane per read and four segments per read: SFS, assembled SFS and extension counts."""
    monkeypatch.setenv("SVDSS_KMER", str(k))
    monkeypatch.setenv("SVDSS_SEGMENTS", str(segments))
    cases = load_golden()
    if k == 16:
        cases = [c for c in cases if c["name"] in ("both_strands_errors", "contig_ends_and_junctions")]
    for case in cases:
        contigs = [from_ascii(c) for c in case["contigs"]]
        ix = svdss_amd.FMDIndex.build(contigs, threads=2).to_device(0)
        assert ix.kmer_k == k
        reads = [from_ascii(r["read"]) for r in case["reads"]]
        flat, offs = svdss_amd.pack_reads(reads)
        raw = _search(ix, flat, offs, False)
        for got, rd, ne in zip(raw.per_read(), case["reads"], raw.n_ext.tolist()):
            assert [list(x) for x in got] == rd["sfs"], case["name"]
            assert ne == rd["n_ext"], case["name"]
        asm = _search(ix, flat, offs, True)
        for got, rd in zip(asm.per_read(), case["reads"]):
            assert [list(x) for x in got] == rd["assembled"], case["name"]
        del ix


@pytest.mark.parametrize("assemble", [False, True])
@pytest.mark.parametrize("seed", [31, 32])
def test_random_reads_match_oracle(assemble, seed):
    ref, hap, svs, flat, offs = small_workload(seed=seed, n_reads=300, read_len=2000)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    got = _search(ix, flat, offs, assemble)
    c, q, l, e = fm.search_batch(flat, offs, assemble)
    assert (got.counts == c).all() and (got.n_ext == e).all()
    assert (got.qs == q).all() and (got.len == l).all()
    assert c.sum() > 0


def test_empty_ragged_and_unaligned():
    ref, hap, svs, flat, offs = small_workload(seed=41, n_reads=10, read_len=300, ref_lens=(40000,))
    reads = [flat[offs[i]:offs[i + 1]] for i in range(10)]
    reads.insert(3, np.zeros(0, np.uint8))
    reads.append(np.zeros(0, np.uint8))
    reads.insert(0, ref[0][5:6])
    flat2, offs2 = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        got = _search(ix, flat2, offs2, assemble)
        c, q, l, e = fm.search_batch(flat2, offs2, assemble)
        assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all() and (got.n_ext == e).all()
    # empty batch
    got = _search(ix, np.zeros(0, np.uint8), np.zeros(1, np.int64), True)
    assert len(got.counts) == 0 and len(got.qs) == 0


def test_record_region_overflow_is_rerun_exactly():
    # an all-N read against an N-free reference yields one SFS per base: far more than the
    # default len/8+8 records per read; the library must rerun with exact capacities
    ref = synth.make_reference([30000], seed=7)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    reads = [np.full(500, 5, np.uint8), ref[0][100:900].copy(), np.full(64, 5, np.uint8),
             synth.revcomp(ref[0][2000:2600])]
    reads[1][400] = 5
    flat, offs = svdss_amd.pack_reads(reads)
    for assemble in (False, True):
        got = _search(ix, flat, offs, assemble)
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        assert got.counts.tolist() == c.tolist() and c[0] == 500
        assert (got.qs == q).all() and (got.len == l).all() and (got.n_ext == e).all()


def test_device_buffers_and_stream():
    import torch
    ref, hap, svs, flat, offs = small_workload(seed=51, n_reads=200, read_len=1500)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    total = int(offs[-1])
    d_reads = torch.zeros(((total + 15) // 16) * 16 + 16, dtype=torch.uint8, device="cuda:0")
    d_reads[:total] = torch.from_numpy(flat).cuda()
    d_offs = torch.from_numpy(offs).cuda()
    pp = svdss_amd.PingPong(ix, assemble=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got = pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), len(offs) - 1, total,
                                         stream=s.cuda_stream)
    c, q, l, e = fm.search_batch(flat, offs, True)
    assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all()
    assert pp.last_total == c.sum() and pp.last_total_ext == e.sum()
    assert pp.last_kernel_ms > 0
    # idempotence: same call, same buffers reused, same answer
    got2 = pp.ping_pong_search_device(d_reads.data_ptr(), d_offs.data_ptr(), len(offs) - 1, total)
    assert (got2.counts == got.counts).all() and (got2.qs == got.qs).all() and (got2.len == got.len).all()


def test_process_batch_putative_filter_and_text():
    # ping_pong.cpp:196-204: XF != 0 reads are skipped when putative; HP is carried as htag
    ref, hap, svs, flat, offs = small_workload(seed=61, n_reads=6, read_len=500, ref_lens=(30000,))
    reads = [flat[offs[i]:offs[i + 1]] for i in range(6)]
    names = [f"r{i}" for i in range(6)]
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    pp = svdss_amd.PingPong(ix)
    sols = pp.process_batch(names, reads, xf=[0, 1, 0, 2, 0, 3], hp=[0, 0, 1, 0, 2, 0])
    assert [s[0] for s in sols] == ["r0", "r2", "r4"] and [s[1] for s in sols] == [0, 1, 2]
    fm = O.OracleFMD.build(ref)
    for (name, hp, sfs), i in zip(sols, (0, 2, 4)):
        raw, _ = fm.ping_pong_search(reads[i])
        assert sfs == O.assemble(raw)
    txt = svdss_amd.output_batch(sols)
    parsed = svdss_amd.parse_sfsfile(txt)
    for name, hp, sfs in sols:
        if sfs:
            assert parsed[name] == [(q, l, hp) for q, l in sfs]
    pp2 = svdss_amd.PingPong(ix, putative=False)
    assert len(pp2.process_batch(names, reads, xf=[0, 1, 0, 2, 0, 3])) == 6


def test_full_size_properties():
    """BASELINE-shape reads (15 kb, 0.5 % errors) at a size the oracle cannot check in
    seconds: size-independent properties of the SFS set."""
    L = 15000
    ref = synth.make_reference([8_000_000], seed=71)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    hap, svs = synth.implant_svs(ref, 40, seed=72)
    flat, offs, truth = synth.simulate_reads(hap, 1500, L, 0.005, seed=73)
    # reads copied verbatim from either strand of the reference: no SFS, exactly L-1 extensions
    rng = np.random.default_rng(74)
    exact = []
    for k in range(64):
        s = int(rng.integers(0, 8_000_000 - L))
        w = ref[0][s:s + L]
        exact.append(synth.revcomp(w) if k % 2 else w.copy())
    eflat, eoffs = svdss_amd.pack_reads(exact)
    got = _search(ix, eflat, eoffs, False)
    assert got.counts.sum() == 0 and (got.n_ext == L - 1).all()

    raw = _search(ix, flat, offs, False)
    asm = _search(ix, flat, offs, True)
    raw2 = _search(ix, flat, offs, False)
    assert (raw.counts == raw2.counts).all() and (raw.qs == raw2.qs).all() and (raw.len == raw2.len).all()
    assert raw.counts.sum() > 1500 * 50
    # extension count == positions consumed by the two loops of ping_pong.cpp:15-22,31-37
    rr, aa = raw.per_read(), asm.per_read()
    for i in range(0, 1500, 7):
        sfs = rr[i]
        qs = [q for q, _ in sfs]
        ends = [q + l for q, l in sfs]
        assert all(a > b for a, b in zip(qs, qs[1:]))        # strictly descending starts
        assert all(a > b for a, b in zip(ends, ends[1:]))    # and ends
        assert aa[i] == O.assemble(sfs)
        a = aa[i]
        assert all(x[0] + x[1] <= y[0] for x, y in zip(a, a[1:]))
    # defining property of an SFS: absent from the reference, both maximal proper substrings present
    chk = 0
    for i in range(0, 1500, 50):
        r = flat[offs[i]:offs[i + 1]]
        for q, l in rr[i][:8]:
            assert ix.count(r[q:q + l]) == 0
            if l > 1:
                assert ix.count(r[q + 1:q + l]) > 0 and ix.count(r[q:q + l - 1]) > 0
            chk += 1
    assert chk > 100
    # a sample against the oracle itself (index contents handed over as a BWT)
    fm = O.OracleFMD.from_bwt(ix.bwt())
    sub = list(range(0, 1500, 100))
    sflat, soffs = svdss_amd.pack_reads([flat[offs[i]:offs[i + 1]] for i in sub])
    c, q, l, e = fm.search_batch(sflat, soffs, False)
    for k, i in enumerate(sub):
        assert raw.counts[i] == c[k] and raw.n_ext[i] == e[k]
    assert split(c, q, l) == [rr[i] for i in sub]


def test_wide_suffix_array_path(monkeypatch):
    """Genomes with >= 2^31 BWT symbols use 64-bit SA entries / intervals (kernel template
    instantiation <uint64_t>); SVDSS_FORCE_SA64 builds that layout for a small input."""
    monkeypatch.setenv("SVDSS_FORCE_SA64", "1")
    ref, hap, svs, flat, offs = small_workload(seed=35, n_reads=150, read_len=1500)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    monkeypatch.delenv("SVDSS_FORCE_SA64")
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        got = _search(ix, flat, offs, assemble)
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all() and (got.n_ext == e).all()


@pytest.mark.parametrize("kmer", ["0", "6", "12"])
def test_kmer_table_orders(monkeypatch, kmer):
    """Any table order K (0 = no table) must give the same SFS; K > log4(n) makes most entries
    'absent with fail depth', K small makes them multi-occurrence intervals."""
    monkeypatch.setenv("SVDSS_KMER", kmer)
    ref, hap, svs, flat, offs = small_workload(seed=36, n_reads=150, read_len=1500)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    assert ix.kmer_k == int(kmer)
    fm = O.OracleFMD.build(ref)
    got = _search(ix, flat, offs, True)
    c, q, l, e = fm.search_batch(flat, offs, True)
    assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all() and (got.n_ext == e).all()


@pytest.mark.parametrize("kmer", [0, 8])
def test_batches_smaller_than_one_fetch(kmer, monkeypatch):
    """A whole batch of fewer than 64 symbols: the kernel fetches 64 bytes of a read at a time, so such a batch is searched
    in a padded copy (csrc/sfs_search.hip) -- one read of 1 .. 63 symbols, several tiny reads, only empty reads; with and
    without the k-mer table."""
    ref, hap, svs, flat, offs = small_workload(seed=43, n_reads=2, read_len=300, ref_lens=(20000,))
    fm = O.OracleFMD.build(ref)
    cases = [[ref[0][100:100 + n].copy()] for n in (1, 2, 15, 16, 17, 31, 47, 63)]
    cases += [[ref[0][7:20].copy(), np.zeros(0, np.uint8), hap[0][50:61].copy(), ref[0][900:925].copy()], [np.zeros(0, np.uint8)] * 3]
    monkeypatch.setenv("SVDSS_KMER", str(kmer))
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    for reads in cases:
        flat2, offs2 = svdss_amd.pack_reads(reads)
        for assemble in (False, True):
            got = _search(ix, flat2, offs2, assemble)
            c, q, l, e = fm.search_batch(flat2, offs2, assemble)
            assert (got.counts == c).all() and (got.qs == q).all() and (got.len == l).all() and (got.n_ext == e).all()


def test_device_results_alias_and_single_rank_gather():
    """bench.py --gpus N hands the library's HBM result buffers to torch.distributed without a
    host hop; check the aliasing tensors and run the gather with a 1-rank NCCL(RCCL) group."""
    import os
    import torch
    import torch.distributed as dist
    from svdss_amd import multi
    ref, hap, svs, flat, offs = small_workload(seed=52, n_reads=120, read_len=1500)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    pp = svdss_amd.PingPong(ix, assemble=True)
    got = pp.ping_pong_search(flat, offs)
    counts, qs, ln = pp.device_results()
    assert counts.is_cuda and counts.dtype == torch.int64 and qs.dtype == torch.int32
    assert (counts.cpu().numpy() == got.counts).all()
    assert (qs.cpu().numpy() == got.qs).all() and (ln.cpu().numpy() == got.len).all()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c2, q2, l2 = multi.gather_sfs(counts, qs, ln)
        torch.cuda.synchronize()
        assert (c2.cpu().numpy() == got.counts).all()
        assert (q2.cpu().numpy() == got.qs).all() and (l2.cpu().numpy() == got.len).all()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("segments", ["1", "2", "4", "16"])
def test_segmented_search_is_exact(monkeypatch, segments):
    """Small batches use several lanes per read; the stitched chains must equal the one-lane result
    (SFS, order, assembled form, extension counts), including reads that fall back."""
    monkeypatch.setenv("SVDSS_SEGMENTS", segments)
    ref, hap, svs, flat, offs = small_workload(seed=38, n_reads=200, read_len=6000, ref_lens=(400000,))
    reads = [flat[offs[i]:offs[i + 1]] for i in range(200)]
    reads += [np.full(1500, 5, np.uint8), ref[0][:5000].copy(), np.zeros(0, np.uint8), ref[0][7:300].copy()]
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        pp = svdss_amd.PingPong(ix, assemble=assemble)
        got = pp.ping_pong_search(flat, offs)
        assert pp.last_segments == int(segments)
        if segments != "1":
            assert 0 < pp.last_fallbacks < 20     # the all-N read cannot be stitched within the overrun
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        assert (got.counts == c).all() and (got.n_ext == e).all()
        assert (got.qs == q).all() and (got.len == l).all()
        pp.close()


@pytest.mark.parametrize("n_reads,direct", [(24, None), (150, None), (150, "0"), (24, "1000000")])
def test_reads_with_long_novel_insertions_in_segmented_launches(monkeypatch, n_reads, direct):
    """What `SVDSS search` sees after `SVDSS smooth`: reads that equal the reference except for one long insertion of
    novel sequence -- an SFS at nearly every base of it.  The segment that owns the insertion overflows its record
    region: it stops at once (second session of round 5; it used to walk the whole stretch for nothing), the read goes
    to the next level -- at most SVDSS_FALLBACK_DIRECT (64) such reads straight to one lane each, more through a level
    with a quarter of the segments first.  Whatever the route: the oracle's SFS, order and extension counts."""
    monkeypatch.setenv("SVDSS_SEGMENTS", "8")
    if direct is not None:
        monkeypatch.setenv("SVDSS_FALLBACK_DIRECT", direct)
    rng = np.random.default_rng(77)
    ref = synth.make_reference([500000], seed=5)
    reads = []
    for k in range(n_reads):
        a = int(rng.integers(0, 500000 - 9000))
        r = ref[0][a:a + 8000].copy()
        ins = rng.integers(1, 5, size=int(rng.integers(300, 2001))).astype(np.uint8)
        cut = int(rng.integers(500, 7500))
        r = np.concatenate([r[:cut], ins, r[cut:]]) if k % 5 else r            # (every fifth read: a plain copy)
        if k % 7 == 3:
            r = synth.revcomp(r)
        reads.append(r)
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        pp = svdss_amd.PingPong(ix, assemble=assemble)
        got = pp.ping_pong_search(flat, offs)
        assert pp.last_segments == 8
        assert pp.last_fallbacks >= n_reads // 2           # the reads with an insertion overflow a segment's region
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        assert (got.counts == c).all() and (got.n_ext == e).all()
        assert (got.qs == q).all() and (got.len == l).all()
        if not assemble:
            assert int(c.max()) > 300                       # an SFS per base of the longest insertions
        pp.close()


@pytest.mark.parametrize("env", [
    {"SVDSS_ORDER": "0"},
    {"SVDSS_ORDER": "1", "SVDSS_TICKETS": "1"},
    {"SVDSS_ORDER": "1", "SVDSS_TICKETS": "64", "SVDSS_BLOCKS": "8"},
    {"SVDSS_SEGMENTS": "4", "SVDSS_BLOCKS": "16", "SVDSS_TICKETS": "3"},
    {"SVDSS_SEGMENTS": "1", "SVDSS_BLOCKS": "4"},
    {"SVDSS_TABLE_FORWARD": "0", "SVDSS_SEGMENTS": "4"},     # k-mer table without the forward-phase outcome
    {"SVDSS_TABLE_FORWARD": "0", "SVDSS_KMER": "9"},
    {"SVDSS_KMER": "9", "SVDSS_SEGMENTS": "2"},              # 4^9 << text: nearly every 9-mer occurs
])
def test_scheduling_knobs_never_change_results(monkeypatch, env):
    """Heavy-reads-first order, per-wavefront ticket pools, resident blocks and segment counts decide when and where a
    read is searched, never what is found: every combination must equal the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref = synth.make_reference([300000], seed=71, repeat_frac=0.2, divergence=0.01)   # plenty of 2-copy repeats
    rng = np.random.default_rng(72)
    reads = []
    for i in range(1400):
        ln = int(rng.integers(300, 2600))
        a = int(rng.integers(0, len(ref[0]) - ln))
        r = ref[0][a:a + ln].copy()
        e = rng.random(ln) < 0.005
        r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        reads.append(r.astype(np.uint8))
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    pp = svdss_amd.PingPong(ix, assemble=True)
    got = pp.ping_pong_search(flat, offs)
    c, q, l, e = fm.search_batch(flat, offs, True)
    assert (got.counts == c).all() and (got.n_ext == e).all()
    assert (got.qs == q).all() and (got.len == l).all()
    pp.close()


@pytest.mark.parametrize("wide", [False, True])
def test_low_copy_repeats_with_n_runs(monkeypatch, wide):
    """SET mode (2-4 occurrences followed in the text) on a reference made mostly of overlapping copies, with N runs in
    the reference and N in the reads, on both suffix-array widths; and the same with SET mode off."""
    if wide:
        monkeypatch.setenv("SVDSS_FORCE_SA64", "1")
    ref = synth.make_reference([120000, 60000], seed=91, repeat_frac=0.6, divergence=0.004, n_runs=(300, 40))
    rng = np.random.default_rng(92)
    reads = []
    for i in range(1200):
        c = ref[int(rng.integers(0, 2))]
        ln = int(rng.integers(200, 3000))
        a = int(rng.integers(0, len(c) - ln))
        r = c[a:a + ln].copy()
        e = rng.random(ln) < 0.004
        r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        if i % 17 == 0:
            r[int(rng.integers(0, ln))] = 5
        reads.append(r.astype(np.uint8))
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    c, q, l, e = fm.search_batch(flat, offs, True)
    for set_mode in ("1", "0"):
        monkeypatch.setenv("SVDSS_SET", set_mode)
        pp = svdss_amd.PingPong(ix, assemble=True)
        got = pp.ping_pong_search(flat, offs)
        assert (got.counts == c).all() and (got.n_ext == e).all()
        assert (got.qs == q).all() and (got.len == l).all()
        pp.close()


def test_binary_search_kernel_is_chosen_by_the_reference(monkeypatch):
    """Round 5: the BS instantiation (deep backward phases finished by binary search of the suffix array) is launched when the
    reference is rich in young repeat families (svdss_index_deep_frac >= 0.35), not for an iid one; SVDSS_BS=0|1 overrides;
    every choice, on one-lane-per-read and segmented launches, gives the oracle's SFS and extension counts."""
    fam = synth.make_family_reference([400000, 200000], seed=7, repeat_frac=0.6, divergence=0.01, n_families=6)
    iid = synth.make_reference([300000], seed=8)
    rng = np.random.default_rng(9)
    for ref, expect_bs in ((fam, True), (iid, False)):
        reads = []
        for i in range(600):
            c = ref[int(rng.integers(0, len(ref)))]
            ln = int(rng.integers(1500, 6000))
            a = int(rng.integers(0, len(c) - ln))
            r = c[a:a + ln].copy()
            e = rng.random(ln) < 0.005
            r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
            reads.append(r.astype(np.uint8))
        flat, offs = svdss_amd.pack_reads(reads)
        monkeypatch.setenv("SVDSS_KMER", "10")     # (a table order at which the small reference has deep k-mers at all)
        ix = svdss_amd.FMDIndex.build(ref).to_device(0)
        assert (ix.deep_frac >= 0.35) == expect_bs, ix.deep_frac
        fm = O.OracleFMD.build(ref)
        c, q, l, e = fm.search_batch(flat, offs, False)
        for bs_env, segs in ((None, "1"), (None, "8"), ("0", "8"), ("1", "1"), ("1", "8")):
            if bs_env is None:
                monkeypatch.delenv("SVDSS_BS", raising=False)
            else:
                monkeypatch.setenv("SVDSS_BS", bs_env)
            monkeypatch.setenv("SVDSS_SEGMENTS", segs)
            pp = svdss_amd.PingPong(ix, assemble=False)
            got = pp.ping_pong_search(flat, offs)
            assert pp.last_used_bs == (expect_bs if bs_env is None else bs_env == "1")
            assert pp.last_segments == int(segs)
            assert (got.counts == c).all() and (got.n_ext == e).all()
            assert (got.qs == q).all() and (got.len == l).all()
            pp.close()
        del ix


def test_gpu_suffix_sorter_builds_the_same_index(monkeypatch, tmp_path):
    """`SVDSS index` sorts suffixes on the GPU when there is one (csrc/index_gpu.hip); the index file must be the one
    the host builder writes (the suffix array of a text is unique), including long repeats, N runs and several contigs."""
    ref = synth.make_reference([180000, 90000, 700], seed=33, repeat_frac=0.5, divergence=0.0005, n_runs=(500, 30))
    ref.append(np.tile(np.array([1, 2, 3, 4], np.uint8), 3000))       # a tandem repeat: many doubling rounds
    a = svdss_amd.FMDIndex.build(ref)
    a.save(str(tmp_path / "gpu.fmd"))
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    b = svdss_amd.FMDIndex.build(ref)
    b.save(str(tmp_path / "cpu.fmd"))
    assert (a.bwt() == b.bwt()).all()
    assert (tmp_path / "gpu.fmd").read_bytes() == (tmp_path / "cpu.fmd").read_bytes()


@pytest.mark.parametrize("kmer", [None, "4", "16"])
def test_degenerate_references_and_reads(monkeypatch, kmer):
    """References the synthetic genomes never produce: records of 1, 2, K-1, K and K+1 bases, a record of N only, a
    record that is one base repeated, a record equal to another's reverse complement (every suffix tied with one of the
    other strand), beside ordinary ones -- index built in HBM (or by the fallback), verified against its text, searched
    by both launch shapes and compared with the oracle; reads of 1 .. K+1 bases, reads of N only, reads that ARE a
    record, its reverse complement, a record plus one base on either side (ping_pong.cpp:4-49 on its boundary cases:
    the phase that starts at the last symbol, the forward phase that runs off the read end at P[l] = 0)."""
    if kmer:
        monkeypatch.setenv("SVDSS_KMER", kmer)
    rng = np.random.default_rng(77)
    base = synth.make_reference([40000, 9000], seed=78, n_runs=(300,))
    tiny = [np.array(x, np.uint8) for x in ([1], [3, 2], [4] * 15, [2, 1, 4, 3] * 4, [1, 2, 3, 4, 1, 1, 2, 2, 3, 3, 4, 4, 2, 4, 1, 3, 2])]
    ref = base + tiny + [np.full(700, 5, np.uint8), np.full(3000, 2, np.uint8), synth.revcomp(base[1][1000:6000])]
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    v = ix.verify()
    assert v["rows"] == ix.size and v["first_bad"] == -1 and v["bad_order"] == v["bad_bwt"] == v["bad_block"] == v["bad_dollar"] == 0, v
    fm = O.OracleFMD.build(ref)
    reads = [np.array([c], np.uint8) for c in (1, 2, 3, 4, 5)]
    reads += [rng.integers(1, 5, size=n).astype(np.uint8) for n in (2, 3, 15, 16, 17, 31, 33, 64, 65, 127, 129)]
    reads += [np.full(n, 5, np.uint8) for n in (1, 16, 200)]
    reads += [t.copy() for t in tiny] + [synth.revcomp(t) for t in tiny]
    reads += [np.concatenate([[3], tiny[3]]), np.concatenate([tiny[3], [1]]), np.concatenate([[4], tiny[4], [4]])]
    reads += [base[0][100:3100].copy(), synth.revcomp(base[0][20000:23000]), base[1][990:6010].copy()]
    reads += [np.full(500, 2, np.uint8), np.full(3001, 2, np.uint8), np.full(40, 3, np.uint8)]
    hap, _ = synth.implant_svs(base, 3, seed=79, min_len=50, max_len=200)
    f2, o2, _ = synth.simulate_reads(hap, 40, 2500, 0.01, seed=80)
    reads += [f2[o2[i]:o2[i + 1]] for i in range(40)]
    flat, offs = svdss_amd.pack_reads(reads)
    for assemble in (False, True):
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        for seg in (None, "1"):
            if seg:
                monkeypatch.setenv("SVDSS_SEGMENTS", seg)
            pp = svdss_amd.PingPong(ix, assemble=assemble)
            got = pp.ping_pong_search(flat, offs)
            pp.close()
            if seg:
                monkeypatch.delenv("SVDSS_SEGMENTS")
            assert (got.counts == c).all() and (got.n_ext == e).all(), (assemble, seg)
            assert (got.qs == q).all() and (got.len == l).all(), (assemble, seg)
    # the records themselves and their reverse complements occur: nothing specific in them
    k = [i for i, r in enumerate(reads) if any(len(r) == len(t) and ((r == t).all() or (r == synth.revcomp(t)).all()) for t in tiny)]
    assert len(k) >= 10 and all(c[i] == 0 for i in k)
