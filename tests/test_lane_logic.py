"""The kernel's per-lane code (sfs_core2.h + sym_window.h + fmd_layout.h), run on
the CPU through tests/lane_emulator.cpp, against the oracle and the golden
vectors.  This validates the state machine, the register window over the read
and the streaming assembler without a GPU; the -m gpu tests repeat the same
comparisons through the real kernel."""
import numpy as np
import pytest

import svdss_amd
from tests import emulator_lib as E
from tests import oracle_lib as O
from tests.common import from_ascii, load_golden, small_workload, split


def test_unaligned_offsets_and_empty_reads():
    # reads start at arbitrary byte offsets of the concatenated buffer; some are empty; every way of evaluating
    for K, use_text in [(0, False), (0, True), (7, True)]:
        ref, hap, svs, flat, offs = small_workload(seed=41, n_reads=10, read_len=300, ref_lens=(40000,))
        reads = [flat[offs[i]:offs[i + 1]] for i in range(10)]
        reads.insert(3, np.zeros(0, np.uint8))
        reads.append(np.zeros(0, np.uint8))
        reads.insert(0, ref[0][5:6])
        flat2, offs2 = svdss_amd.pack_reads(reads)
        ix = svdss_amd.FMDIndex.build(ref, threads=2)
        fm = O.OracleFMD.build(ref)
        for assemble in (False, True):
            c, q, l, e, ops = E.search2(ix, flat2, offs2, assemble, K, use_text)
            c2, q2, l2, e2 = fm.search_batch(flat2, offs2, assemble)
            assert (c == c2).all() and (q == q2).all() and (l == l2).all() and (e == e2).all()
        assert c[4] == 0 and c[-1] == 0


def test_batches_smaller_than_one_fetch():
    """A whole batch of fewer than 64 symbols (the kernel fetches 64 bytes of a read at a time and searches such a batch
    in a padded copy, csrc/sfs_search.hip): one read of 1 .. 63 symbols, several tiny reads, only empty reads."""
    ref, hap, svs, flat, offs = small_workload(seed=43, n_reads=2, read_len=300, ref_lens=(20000,))
    ix = svdss_amd.FMDIndex.build(ref, threads=2)
    fm = O.OracleFMD.build(ref)
    cases = [[ref[0][100:100 + n].copy()] for n in (1, 2, 15, 16, 17, 31, 47, 63)]
    cases += [[ref[0][7:20].copy(), np.zeros(0, np.uint8), hap[0][50:61].copy(), ref[0][900:925].copy()], [np.zeros(0, np.uint8)] * 3]
    for reads in cases:
        flat2, offs2 = svdss_amd.pack_reads(reads)
        for K, use_text in [(0, False), (5, True)]:
            for assemble in (False, True):
                c, q, l, e, ops = E.search2(ix, flat2, offs2, assemble, K, use_text)
                c2, q2, l2, e2 = fm.search_batch(flat2, offs2, assemble)
                assert (c == c2).all() and (q == q2).all() and (l == l2).all() and (e == e2).all()


# ---- v2 state machine (sfs_core2.h): k-mer table + LF + unique-match TEXT mode ----

V2_CONFIGS = [(0, False), (0, True), (5, False), (7, True), (10, True)]


def test_v2_golden_through_lane_code():
    for case in load_golden():
        contigs = [from_ascii(c) for c in case["contigs"]]
        ix = svdss_amd.FMDIndex.build(contigs, threads=2)
        reads = [from_ascii(r["read"]) for r in case["reads"]]
        flat, offs = svdss_amd.pack_reads(reads)
        for K, use_text in V2_CONFIGS:
            c, q, l, e, ops = E.search2(ix, flat, offs, False, K, use_text)
            for got, rd, ne in zip(split(c, q, l), case["reads"], e.tolist()):
                assert [list(x) for x in got] == rd["sfs"], (case["name"], K, use_text)
                assert ne == rd["n_ext"]
            c, q, l, e, ops = E.search2(ix, flat, offs, True, K, use_text)
            for got, rd in zip(split(c, q, l), case["reads"]):
                assert [list(x) for x in got] == rd["assembled"]


@pytest.mark.parametrize("assemble", [False, True])
def test_v2_random_reads_match_oracle(assemble):
    ref, hap, svs, flat, offs = small_workload(seed=33, n_reads=40, read_len=1500)
    reads = [flat[offs[i]:offs[i + 1]] for i in range(40)]
    # a short first read (TEXT windows would start before the buffer), N reads, empty read, 1-mers
    reads = [ref[0][3:40].copy()] + reads + [np.full(70, 5, np.uint8), np.zeros(0, np.uint8), ref[1][:1].copy(),
                                              ref[0][0:900].copy(), hap[0][:700].copy()]
    reads[-2][450] = 5
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref, threads=4)
    fm = O.OracleFMD.build(ref)
    c2, q2, l2, e2 = fm.search_batch(flat, offs, assemble)
    for K, use_text in V2_CONFIGS:
        c, q, l, e, ops = E.search2(ix, flat, offs, assemble, K, use_text)
        assert (c == c2).all() and (e == e2).all(), (K, use_text)
        assert (q == q2).all() and (l == l2).all()
        if K:
            assert ops["TABLE"] > 0
        if use_text:
            assert ops["TEXT"] > 0
    # the point of v2: an order of magnitude fewer memory operations than extensions
    assert sum(ops[k] for k in ("LF", "TABLE", "SA", "TEXT", "FILL")) * 4 < e2.sum()


def test_v2_repeats_and_long_unique_stretches():
    # diverged repeats keep intervals > 1 for long (LF mode), exact reads exercise multi-window TEXT runs
    from svdss_amd import synth
    ref = synth.make_reference([90000], seed=77, repeat_frac=0.4, divergence=0.002)
    ix = svdss_amd.FMDIndex.build(ref, threads=4)
    fm = O.OracleFMD.build(ref)
    rng = np.random.default_rng(5)
    reads = []
    for k in range(12):
        s = int(rng.integers(0, 80000))
        w = ref[0][s:s + 3000].copy()
        if k % 3 == 0:
            w[int(rng.integers(0, 3000))] = 5
        reads.append(synth.revcomp(w) if k % 2 else w)
    flat, offs = svdss_amd.pack_reads(reads)
    c2, q2, l2, e2 = fm.search_batch(flat, offs, False)
    for K, use_text in [(8, True), (0, True), (12, True)]:   # (12: 16.7 M table entries; K = 14 / 16 need the GPU, test_sfs_gpu.py)
        c, q, l, e, ops = E.search2(ix, flat, offs, False, K, use_text)
        assert (c == c2).all() and (e == e2).all() and (q == q2).all() and (l == l2).all()


@pytest.mark.parametrize("n_seg", [1, 4])
def test_v2_set_mode_in_low_copy_repeats(n_seg):
    """SET mode (2-4 occurrences followed in the text, 16 symbols per operation) must leave SFS and extension counts
    untouched while replacing most LF steps of reads in low-copy repeats."""
    from svdss_amd import synth
    ref = synth.make_reference([400000], seed=3, repeat_frac=0.3, divergence=0.006, n_runs=(60,))
    rng = np.random.default_rng(5)
    reads = []
    for i in range(40):
        ln = int(rng.integers(1500, 5000))
        a = int(rng.integers(0, len(ref[0]) - ln))
        r = ref[0][a:a + ln].copy()
        e = rng.random(ln) < 0.005
        r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        if i % 7 == 0:
            r[int(rng.integers(0, ln))] = 5
        reads.append(r.astype(np.uint8))
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref)
    fm = O.OracleFMD.build(ref)
    c, q, l, e = fm.search_batch(flat, offs, True)
    ops = {}
    for use_set in (False, True):
        got = E.search2(ix, flat, offs, assemble=True, K=9, n_seg=n_seg, use_set=use_set)
        assert (got[0] == c).all() and (got[3] == e).all()
        assert (got[1] == q).all() and (got[2] == l).all()
        ops[use_set] = got[4]
    assert ops[False]["SET"] == 0 and ops[True]["SET"] > 0 and ops[True]["SA_SET"] > 0
    assert ops[True]["LF"] < ops[False]["LF"] // 2


def test_kmer_table_entries_against_plain_substring_search():
    """Every entry of the k-mer table (sv_table_entry): d = how many trailing symbols of the K-mer occur, and for an
    absent K-mer df = how many leading symbols of the d + 1 that fail together still occur -- the outcome of the
    forward phase the reference starts there (ping_pong.cpp:28-37), which the kernel takes from the entry instead of
    a second lookup.  Checked against substring search in the contigs and their reverse complements."""
    rng = np.random.default_rng(12)
    contigs = [rng.integers(1, 5, size=n).astype(np.uint8) for n in (700, 450)]
    ix = svdss_amd.FMDIndex.build(contigs, threads=2)
    letters = "$ACGTN"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    strands = []
    for c in contigs:
        fw = "".join(letters[x] for x in c)
        strands += [fw, "".join(comp[ch] for ch in reversed(fw))]
    occurs = lambda x: any(x in t for t in strands)
    K = 6
    lo, info = E.kmer_table(ix, K)
    n_empty = n_df = n_ext_checked = 0
    for key in range(1 << (2 * K)):
        w = "".join("ACGT"[(key >> (2 * i)) & 3] for i in range(K))     # first symbol in the low bits
        d = 0
        while d < K and occurs(w[K - 1 - d:]):
            d += 1
        typ = int(info[key]) >> 62
        if d == K:
            assert typ in (1, 2, 3), (w, typ)
            starts = [(t, i) for t in strands for i in range(len(t) - K + 1) if t[i:i + K] == w]
            size = len(starts)
            assert (typ == 1) == (size == 1) and (typ == 3) == (2 <= size <= 4) and (typ == 2) == (size >= 5), (w, typ, size)
            if typ in (1, 3):
                # extension symbols: the 6 text symbols in front of every occurrence, nearest first, '$' (0) from the
                # record start on -- as a multiset (the entry lists the occurrences in suffix-array order)
                want = sorted(sum(("$ACGTN".index(t[i - 1 - e]) if i - 1 - e >= 0 else 0) << (3 * e) for e in range(6))
                              for t, i in starts)
                ilo, iin = int(lo[key]), int(info[key])
                if typ == 1:
                    got = [(iin >> 40) & 0x3ffff]
                else:
                    assert (iin >> 59) & 7 == size
                    got = [((iin >> (18 * j)) & 0x3ffff) if j < 3 else ((ilo >> 36) & 0x3ffff) for j in range(size)]
                assert sorted(got) == want, (w, got, want)
                n_ext_checked += 1
            continue
        n_empty += 1
        assert typ == 0 and (int(info[key]) & 0xff) == d, (w, d, int(info[key]) & 0xff)
        if d == 0:
            continue
        f = w[K - 1 - d:]                      # d + 1 symbols that do not occur together
        df = 0
        while df < len(f) and occurs(f[:df + 1]):
            df += 1
        assert df <= d
        assert ((int(info[key]) >> 8) & 0xff) == df, (w, f, df, (int(info[key]) >> 8) & 0xff)
        n_df += 1
    assert n_empty > 500 and n_df > 500 and n_ext_checked > 500


@pytest.mark.parametrize("seed", [101, 102, 103, 104])
def test_v2_randomised_configurations(seed):
    """Random small genomes (some tiny, so that most K-mers are absent and the table's forward-phase outcome is used on
    nearly every SFS), error rates, N content, table orders and segment counts: SFS and extension counts of the lane
    code equal the oracle's in every configuration."""
    from svdss_amd import synth
    rng = np.random.default_rng(seed)
    ref_len = int(rng.choice([600, 3000, 20000, 120000]))
    ref = synth.make_reference([ref_len, max(300, ref_len // 3)], seed=seed, repeat_frac=float(rng.choice([0.0, 0.2])),
                               n_runs=(int(rng.integers(0, 3)),))
    err = float(rng.choice([0.0, 0.005, 0.03]))
    reads = []
    for i in range(16):
        c = ref[int(rng.integers(0, 2))]
        ln = int(rng.integers(20, min(len(c), 2500)))
        a = int(rng.integers(0, len(c) - ln + 1))
        r = c[a:a + ln].copy()
        e = rng.random(ln) < err
        r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        if i % 5 == 0:
            r[int(rng.integers(0, ln))] = 5
        reads.append(synth.revcomp(r) if i % 2 else r)
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(ref, threads=2)
    fm = O.OracleFMD.build(ref)
    for assemble in (False, True):
        c2, q2, l2, e2 = fm.search_batch(flat, offs, assemble)
        for K in (0, 4, 7, 9, 11):
            for n_seg in (1, 4):
                c, q, l, e, ops = E.search2(ix, flat, offs, assemble, K, True, n_seg=n_seg)
                assert (c == c2).all() and (e == e2).all(), (seed, K, n_seg, assemble)
                assert (q == q2).all() and (l == l2).all(), (seed, K, n_seg, assemble)


def test_lane_code_under_sanitizers():
    """The kernel's per-lane code (sfs_core2.h, sym_window.h, fmd_layout.h: the search state machine, the
    k-mer table entries, SET / TEXT windows) has no GPU sanitizer to run under on this pool; its CPU emulation has:
    this module's tests again in a process with AddressSanitizer preloaded and the emulator built with
    -fsanitize=address,undefined -- an out-of-bounds window index, a shift by the word size, a signed overflow in the
    interval arithmetic would stop it."""
    import os
    import subprocess
    import sys
    if os.environ.get("SVDSS_EMU_SANITIZE") == "1":
        return                                        # (this IS the sanitized run)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside this g++")
    env = dict(os.environ, SVDSS_EMU_SANITIZE="1", LD_PRELOAD=asan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_lane_logic.py", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not under_sanitizers and not 103 and not 104"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("n_seg,bs_after", [(1, 0), (4, 0), (1, 2), (4, 48)])
def test_v2_deep_intervals_by_binary_search(n_seg, bs_after, monkeypatch):
    """BS mode (round 4): a backward phase that starts on a K-mer with 8 or more occurrences -- reads inside repeat
    families -- is finished by binary search of the suffix array among the rows of the reverse-complemented K-mer (text
    comparisons in the other strand) instead of one rank step per symbol.  Families of near-identical copies on both
    strands, several records (the mirror image of a text position depends on its record), N runs, reads that run into the
    start of the read inside a repeat: SFS and extension counts as the oracle's, with most LF steps gone."""
    from svdss_amd import synth
    # (the binary search takes a phase over when more than this many rank steps are still expected after the first eight:
    # 0 = at once; the shipped value is 48 -- intervals that slow are rare at this scale, so the assertions on the
    # operation counts are for 0 / 2 only)
    monkeypatch.setenv("SVDSS_BS_AFTER", str(bs_after))
    rng = np.random.default_rng(17)
    contigs = []
    fam = [rng.integers(1, 5, size=int(rng.integers(300, 1500)), dtype=np.uint8) for _ in range(6)]
    for ci, L in enumerate((60000, 25000, 8000)):
        c = rng.integers(1, 5, size=L, dtype=np.uint8)
        at = 500
        while at + 2000 < L:
            f = fam[int(rng.integers(0, len(fam)))].copy()
            e = rng.random(len(f)) < 0.01                     # 1 % divergence between copies
            f[e] = (f[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
            if rng.random() < 0.5:
                f = synth.revcomp(f)
            c[at:at + len(f)] = f
            at += len(f) + int(rng.integers(100, 900))
        if ci == 1:
            c[3000:3040] = 5
        contigs.append(c)
    reads = []
    for i in range(48):
        c = contigs[i % 3]
        ln = int(rng.integers(400, 3000))
        a = int(rng.integers(0, len(c) - ln))
        r = c[a:a + ln].copy()
        e = rng.random(ln) < 0.005
        r[e] = (r[e] - 1 + rng.integers(1, 4, size=int(e.sum()))) % 4 + 1
        if i % 2:
            r = synth.revcomp(r)
        reads.append(r.astype(np.uint8))
    reads.append(fam[0][:200].copy())                        # a read that is nothing but the start of a family
    reads.append(synth.revcomp(fam[1])[-300:].copy())
    flat, offs = svdss_amd.pack_reads(reads)
    ix = svdss_amd.FMDIndex.build(contigs)
    fm = O.OracleFMD.build(contigs)
    for assemble in (False, True):
        c, q, l, e = fm.search_batch(flat, offs, assemble)
        ops = {}
        for K in (6, 8):
            for use_bs in (False, True):
                got = E.search2(ix, flat, offs, assemble=assemble, K=K, n_seg=n_seg, use_bs=use_bs)
                assert (got[0] == c).all() and (got[3] == e).all(), (K, use_bs)
                assert (got[1] == q).all() and (got[2] == l).all(), (K, use_bs)
                ops[(K, use_bs)] = got[4]
        assert ops[(6, False)]["BS_SA"] == 0
        if bs_after == 0:
            assert ops[(6, True)]["BS_SA"] > 100 and ops[(6, True)]["LF"] < ops[(6, False)]["LF"] // 3
        if bs_after == 2:
            assert ops[(6, True)]["BS_SA"] > 10 and ops[(6, True)]["LF"] < ops[(6, False)]["LF"]
