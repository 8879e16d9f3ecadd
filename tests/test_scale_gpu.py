"""Parity at the sizes and table orders the product actually runs: K = 14 / 16 tables checked
bit for bit against the oracle, BASELINE config 2 (chr20 length, 3x), a reference above 2^31 BWT symbols (64-bit
suffix array and intervals where 32 bits would overflow), BASELINE config 4's reference (GRCh38 primary lengths, above
2^32 symbols), and the index builder of csrc/index_gpu.hip against the
host builder.  Integer work: every comparison is bit-exact."""
import os

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from tests import oracle_lib as O
from tests.common import small_workload, split

pytestmark = pytest.mark.gpu


def _search(ix, flat, offs, assemble):
    pp = svdss_amd.PingPong(ix, assemble=assemble)
    b = pp.ping_pong_search(flat, offs)
    segs, fb = pp.last_segments, pp.last_fallbacks
    pp.close()
    return b, segs, fb


def _verified(ix):
    """The index against its own text by direct comparison (csrc/index_verify.hip, independent of the builders): only
    after this is an oracle built from the index's BWT an independent checker."""
    v = ix.verify()
    assert v["rows"] == ix.size and v["first_bad"] == -1, v
    assert v["bad_order"] == v["bad_bwt"] == v["bad_range"] == v["bad_block"] == v["bad_dollar"] == 0, v
    return v


def _same(got, c, q, l, e):
    assert (got.counts == c).all() and (got.n_ext == e).all()
    assert (got.qs == q).all() and (got.len == l).all()


@pytest.mark.parametrize("kmer", ["14", "16"])
def test_large_table_orders_match_oracle(monkeypatch, kmer):
    """The table orders of chr20 / GRCh38 runs (K = 14..16) on a reference small enough for the full oracle: at
    K = 16 the `K >= 16` masks of sv_ring_kmer / sv_key_revcomp, the shift-by-0 path and the 64 GiB table build run;
    1,200 reads, so both the several-lanes-per-read launch (with heavy-first ordering) and the one-lane launch."""
    monkeypatch.setenv("SVDSS_KMER", kmer)
    ref, hap, svs, flat, offs = small_workload(seed=61, ref_lens=(300000, 120000), n_reads=1200, read_len=3000)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    assert ix.kmer_k == int(kmer)
    fm = O.OracleFMD.build(ref)
    for assemble in (True, False):
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        got, segs, _ = _search(ix, flat, offs, assemble)
        assert segs > 1                       # small batch: segmented launch
        _same(got, c, q, l, e)
        monkeypatch.setenv("SVDSS_SEGMENTS", "1")
        got1, segs1, _ = _search(ix, flat, offs, assemble)
        monkeypatch.delenv("SVDSS_SEGMENTS")
        assert segs1 == 1
        _same(got1, c, q, l, e)
    assert c.sum() > 1200 * 5
    # reads with N and reads shorter than K
    extra = [np.full(40, 5, np.uint8), flat[offs[3]:offs[3] + 9].copy(), flat[offs[5]:offs[5] + 16].copy(),
             np.concatenate([flat[offs[7]:offs[7] + 500], np.full(3, 5, np.uint8), flat[offs[7] + 500:offs[7] + 900]])]
    eflat, eoffs = svdss_amd.pack_reads(extra)
    c, q, l, e = fm.search_batch(eflat, eoffs, False)
    got, _, _ = _search(ix, eflat, eoffs, False)
    _same(got, c, q, l, e)


def test_config2_chr20_3x():
    """BASELINE config 2: chr20-length contig (64,444,167 bp), 12,889 reads of 15 kb (3x), K = 16 chosen by the
    library itself.  A sample of the reads against the oracle (index contents handed over as a BWT), every read
    against the one-lane launch, and the size-independent properties of an SFS set on all of them."""
    L, n_reads = 15000, 12889
    ref = synth.make_reference([64_444_167], seed=11)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    assert ix.kmer_k == 16 and ix.size == 2 * (64_444_167 + 1)
    hap, svs = synth.implant_svs(ref, 64, seed=12)
    flat, offs, truth = synth.simulate_reads(hap, n_reads, L, 0.005, seed=13)
    raw, segs, _ = _search(ix, flat, offs, False)
    asm, _, _ = _search(ix, flat, offs, True)
    assert segs > 1
    os.environ["SVDSS_SEGMENTS"] = "1"
    try:
        raw1, segs1, _ = _search(ix, flat, offs, False)
    finally:
        del os.environ["SVDSS_SEGMENTS"]
    assert segs1 == 1
    _same(raw1, raw.counts, raw.qs, raw.len, raw.n_ext)
    # oracle on a sample (every 16th read: ~800 reads)
    fm = O.OracleFMD.from_bwt(ix.bwt())
    sub = list(range(0, n_reads, 16))
    sflat, soffs = svdss_amd.pack_reads([flat[offs[i]:offs[i + 1]] for i in sub])
    rr, aa = raw.per_read(), asm.per_read()
    c, q, l, e = fm.search_batch(sflat, soffs, False)
    assert (raw.counts[sub] == c).all() and (raw.n_ext[sub] == e).all()
    assert split(c, q, l) == [rr[i] for i in sub]
    c, q, l, e = fm.search_batch(sflat, soffs, True)
    assert (asm.counts[sub] == c).all()
    assert split(c, q, l) == [aa[i] for i in sub]
    # properties on every read: strictly descending starts and ends, assembly = oracle's assemble of the raw list
    qs, ln, cnt = raw.qs.astype(np.int64), raw.len.astype(np.int64), raw.counts
    rid = np.repeat(np.arange(n_reads), cnt)
    same = rid[1:] == rid[:-1]                       # adjacent records of one read
    assert (qs[1:][same] < qs[:-1][same]).all()
    assert ((qs + ln)[1:][same] < (qs + ln)[:-1][same]).all()
    assert (qs >= 0).all() and (qs + ln <= L).all() and (ln > 0).all()
    for i in range(0, n_reads, 97):
        assert aa[i] == O.assemble(rr[i])
    # reads that cover an implanted SV breakpoint carry an SFS over it (truth recovery at the read level)
    assert raw.counts.sum() > n_reads * 300


def test_reference_above_2_31_symbols():
    """A 1.1 Gb contig: 2.2e9 BWT symbols, past the 32-bit range -- the <uint64_t> kernel instantiation and the
    64-bit suffix array run where 32 bits would overflow (SVDSS_FORCE_SA64 only ever exercised them at 230 kb).
    Built in HBM by csrc/index_gpu.hip (several pieces), checked against the oracle on a sample of the reads."""
    L, n_reads = 15000, 2048
    n_ref = 1_100_000_000
    ref = synth.make_reference([n_ref], seed=81)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    assert ix.size == 2 * (n_ref + 1) > 2 ** 31
    _verified(ix)
    # reads from the far end of the contig, so that text positions and SA indices above 2^31 and 2^32 are hit
    tail = [ref[0][n_ref - 40_000_000:]]
    hap, svs = synth.implant_svs(tail, 8, seed=82)
    flat, offs, truth = synth.simulate_reads(hap, n_reads, L, 0.005, seed=83)
    raw, _, _ = _search(ix, flat, offs, False)
    asm, _, _ = _search(ix, flat, offs, True)
    exact = [ref[0][n_ref - 20000:n_ref - 5000].copy(), synth.revcomp(ref[0][n_ref - 20000:n_ref - 5000]),
             ref[0][:L].copy()]
    eflat, eoffs = svdss_amd.pack_reads(exact)
    got, _, _ = _search(ix, eflat, eoffs, False)
    assert got.counts.sum() == 0 and (got.n_ext == L - 1).all()
    fm = O.OracleFMD.from_bwt(ix.bwt())
    sub = list(range(0, n_reads, 8))
    sflat, soffs = svdss_amd.pack_reads([flat[offs[i]:offs[i + 1]] for i in sub])
    rr, aa = raw.per_read(), asm.per_read()
    c, q, l, e = fm.search_batch(sflat, soffs, False)
    assert (raw.counts[sub] == c).all() and (raw.n_ext[sub] == e).all()
    assert split(c, q, l) == [rr[i] for i in sub]
    c, q, l, e = fm.search_batch(sflat, soffs, True)
    assert split(c, q, l) == [aa[i] for i in sub]
    assert raw.counts.sum() > n_reads * 300


def test_config4_grch38_primary_lengths():
    """BASELINE config 4 at its full reference size: 24 contigs with the GRCh38 primary lengths (3,088,269,832 bp,
    6.18e9 BWT symbols -- text positions and suffix-array rows beyond 2^32, the K = 16 table with three quarters of all
    16-mers present, UNIQUE / FEW entries with extension symbols on nearly every lookup), index built in HBM the way
    bench.py builds it.  15 kb reads with 0.5 % errors from the first, a middle and the last contig: the several-lanes-
    per-read launch and the one-lane launch against each other on all reads, a sample against the oracle (index
    contents handed over as a BWT), exact reads of both strands, and the size-independent properties of an SFS set."""
    import bench
    L, n_reads = 15000, 2048
    lens = list(bench.GRCH38_PRIMARY)
    ref = synth.make_reference(lens, seed=11)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    assert ix.kmer_k == 16 and ix.size == 2 * (sum(lens) + len(lens)) > 2 ** 32
    _verified(ix)
    span = 30_000_000
    pieces = [ref[0][:span], ref[11][5_000_000:5_000_000 + span], ref[23][len(ref[23]) - span:]]
    hap, svs = synth.implant_svs(pieces, 12, seed=42)
    flat, offs, truth = synth.simulate_reads(hap, n_reads, L, 0.005, seed=43)
    raw, segs, _ = _search(ix, flat, offs, False)
    asm, _, _ = _search(ix, flat, offs, True)
    assert segs > 1
    os.environ["SVDSS_SEGMENTS"] = "1"
    try:
        raw1, segs1, _ = _search(ix, flat, offs, False)
        asm1, _, _ = _search(ix, flat, offs, True)
    finally:
        del os.environ["SVDSS_SEGMENTS"]
    assert segs1 == 1
    _same(raw1, raw.counts, raw.qs, raw.len, raw.n_ext)
    _same(asm1, asm.counts, asm.qs, asm.len, asm.n_ext)
    last = ref[23]
    exact = [last[len(last) - 20000:len(last) - 5000].copy(), synth.revcomp(last[len(last) - 20000:len(last) - 5000]),
             ref[0][:L].copy(), synth.revcomp(ref[12][7_000_000:7_000_000 + L])]
    eflat, eoffs = svdss_amd.pack_reads(exact)
    got, _, _ = _search(ix, eflat, eoffs, False)
    assert got.counts.sum() == 0 and (got.n_ext == L - 1).all()
    fm = O.OracleFMD.from_bwt(ix.bwt())
    sub = list(range(0, n_reads, 8))
    sflat, soffs = svdss_amd.pack_reads([flat[offs[i]:offs[i + 1]] for i in sub])
    rr, aa = raw.per_read(), asm.per_read()
    c, q, l, e = fm.search_batch(sflat, soffs, False)
    assert (raw.counts[sub] == c).all() and (raw.n_ext[sub] == e).all()
    assert split(c, q, l) == [rr[i] for i in sub]
    c, q, l, e = fm.search_batch(sflat, soffs, True)
    assert split(c, q, l) == [aa[i] for i in sub]
    qs, ln, cnt = raw.qs.astype(np.int64), raw.len.astype(np.int64), raw.counts
    rid = np.repeat(np.arange(n_reads), cnt)
    same = rid[1:] == rid[:-1]
    assert (qs[1:][same] < qs[:-1][same]).all() and ((qs + ln)[1:][same] < (qs + ln)[:-1][same]).all()
    assert (qs >= 0).all() and (qs + ln <= L).all() and (ln > 0).all()
    for i in range(0, n_reads, 61):
        assert aa[i] == O.assemble(rr[i])
    assert raw.counts.sum() > n_reads * 300


def test_repeat_rich_reference_above_2_32_symbols():
    """GRCh38 primary lengths with 45 % of the bases in copies of 40 repeat families (thousands of near-identical
    copies each; contigs 1-12 at 1 % divergence, 13-24 at 5 %): the LF / SET paths on intervals of many occurrences
    with 64-bit bounds, which the iid references above barely touch (ping_pong.cpp:15-22 on deep intervals).  Reads
    are drawn from stretches that are half repeats; a sample goes through the oracle (built from the BWT of the index,
    which is verified row by row first), all reads through both launch shapes."""
    import bench
    L, n_reads = 15000, 1536
    lens = list(bench.GRCH38_PRIMARY)
    ref = synth.make_family_reference(lens[:12], seed=101, divergence=0.01) + \
        synth.make_family_reference(lens[12:], seed=102, divergence=0.05)
    ix = svdss_amd.FMDIndex.build(ref, device=0)
    assert ix.size == 2 * (sum(lens) + len(lens)) > 2 ** 32
    v = _verified(ix)
    assert v["max_lcp"] >= 300                     # identical stretches of family copies
    span = 20_000_000
    pieces = [ref[0][1_000_000:1_000_000 + span], ref[23][len(ref[23]) - span:]]
    hap, svs = synth.implant_svs(pieces, 8, seed=103)
    flat, offs, truth = synth.simulate_reads(hap, n_reads, L, 0.005, seed=104)
    raw, segs, _ = _search(ix, flat, offs, False)
    asm, _, _ = _search(ix, flat, offs, True)
    os.environ["SVDSS_SEGMENTS"] = "1"
    try:
        raw1, segs1, _ = _search(ix, flat, offs, False)
    finally:
        del os.environ["SVDSS_SEGMENTS"]
    assert segs > 1 and segs1 == 1
    _same(raw1, raw.counts, raw.qs, raw.len, raw.n_ext)
    fm = O.OracleFMD.from_bwt(ix.bwt())
    sub = list(range(0, n_reads, 6))
    sflat, soffs = svdss_amd.pack_reads([flat[offs[i]:offs[i + 1]] for i in sub])
    rr, aa = raw.per_read(), asm.per_read()
    c, q, l, e = fm.search_batch(sflat, soffs, False)
    assert (raw.counts[sub] == c).all() and (raw.n_ext[sub] == e).all()
    assert split(c, q, l) == [rr[i] for i in sub]
    c, q, l, e = fm.search_batch(sflat, soffs, True)
    assert split(c, q, l) == [aa[i] for i in sub]
    # repeats make SFS rarer and longer than on iid sequence, but every read still carries its errors' SFS
    assert raw.counts.sum() > n_reads * 100
    # exact reads from inside a 1 %-family region on both strands: nothing specific
    exact = [ref[0][5_000_000:5_000_000 + L].copy(), synth.revcomp(ref[3][7_000_000:7_000_000 + L])]
    eflat, eoffs = svdss_amd.pack_reads(exact)
    got, _, _ = _search(ix, eflat, eoffs, False)
    assert got.counts.sum() == 0 and (got.n_ext == L - 1).all()


@pytest.mark.parametrize("piece", [None, "100000", "20000"])
@pytest.mark.parametrize("wide", [False, True])
def test_index_built_in_hbm_is_the_host_builders_index(monkeypatch, tmp_path, piece, wide):
    """svdss_index_build_device (csrc/index_gpu.hip: text, bucketed key sort, prefix doubling on the tied suffixes,
    BWT blocks from wave ballots) writes the file the host builder writes, for one piece and for many small ones,
    32- and 64-bit suffix arrays; long repeats, N runs, several contigs, a tandem repeat."""
    ref = synth.make_reference([180000, 90000, 700], seed=33, repeat_frac=0.5, divergence=0.0005, n_runs=(500, 30))
    ref.append(np.tile(np.array([1, 2, 3, 4], np.uint8), 3000))
    if wide:
        monkeypatch.setenv("SVDSS_FORCE_SA64", "1")
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    svdss_amd.FMDIndex.build(ref).save(str(tmp_path / "cpu.fmd"))
    monkeypatch.delenv("SVDSS_INDEX_CPU")
    if piece:
        monkeypatch.setenv("SVDSS_SA_PIECE", piece)
    g = svdss_amd.FMDIndex.build(ref, device=0)
    g.save(str(tmp_path / "gpu.fmd"))
    assert (tmp_path / "gpu.fmd").read_bytes() == (tmp_path / "cpu.fmd").read_bytes()
    # and it is searchable as built (resident, k-mer table included)
    flat, offs, _ = synth.simulate_reads(ref[:2], 64, 1500, 0.005, seed=5)
    fm = O.OracleFMD.build(ref)
    got, _, _ = _search(g, flat, offs, True)
    _same(got, *fm.search_batch(flat, offs, True))


def test_degenerate_text_falls_back_to_the_host_builder(monkeypatch, tmp_path):
    """One 4-symbol bucket larger than a piece (poly-A) is refused by the GPU sorter; svdss_index_build_device then
    builds on the host and uploads -- same index, no error."""
    ref = [np.full(5000, 1, np.uint8), synth.make_reference([3000], seed=2)[0]]
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    svdss_amd.FMDIndex.build(ref).save(str(tmp_path / "cpu.fmd"))
    monkeypatch.delenv("SVDSS_INDEX_CPU")
    monkeypatch.setenv("SVDSS_SA_PIECE", "3000")
    g = svdss_amd.FMDIndex.build(ref, device=0)
    g.save(str(tmp_path / "gpu.fmd"))
    assert (tmp_path / "gpu.fmd").read_bytes() == (tmp_path / "cpu.fmd").read_bytes()
    assert g.kmer_k > 0
