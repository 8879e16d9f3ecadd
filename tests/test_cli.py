"""`SVDSS` host CLI: process boundary of SURVEY 8(b) B1 (flags, exit codes, stdout text)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import svdss_amd
from svdss_amd import synth
from tests import bam_writer, oracle_lib as O
from tests.common import ROOT, small_workload

from tests.common import BIN  # noqa: E402


def run(*args, **kw):
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=600, **kw)


def test_version_usage_and_errors():
    r = run("--version")
    assert r.returncode == 0 and r.stdout == "SVDSS, v2.1.1\n"          # main.cpp:45-47
    r = run()
    assert r.returncode == 1 and "Usage" in r.stderr and r.stdout == ""  # main.cpp:27-31
    r = run("frobnicate")
    assert r.returncode == 1                                             # main.cpp:78-80
    r = run("search", "--index", "x.fmd")
    assert r.returncode == 1 and "Usage: SVDSS search" in r.stderr       # main.cpp:63-66
    r = run("search", "--index", "/nonexistent.fmd", "--fastx", "/nonexistent.fq")
    assert r.returncode == 1 and "critical" in r.stderr                  # message + exit(1)
    r = run("call", "--reference", "a", "--bam", "b", "--sfs", "c")
    assert r.returncode == 1
    # --clipped is declared by the reference (config.cpp:46, EXPERIMENTAL) and taken (tests/test_clipped.py)
    r = run("call", "--reference", "a", "--bam", "b", "--sfs", "c", "--clipped")
    assert r.returncode == 1 and "does not exist" not in r.stderr
    r = run("call", "--reference", "a", "--bam", "b", "--sfs", "c", "--frobnicate")
    assert r.returncode == 1 and "does not exist" in r.stderr
    # --help prints the mode's own usage and succeeds (main.cpp:47-50, config.cpp:12-24); cxxopts' rules for the rest
    # (tests/test_ref_pins.py holds the parser against the reference's): positional arguments are left alone, a number
    # that is not one is an error
    for mode in ("search", "call", "smooth"):
        r = run(mode, "--help")
        assert r.returncode == 0 and f"Usage: SVDSS {mode}" in r.stderr and r.stdout == ""
    r = run("search", "stray", "--index", "/nonexistent.fmd", "--fastx", "/nonexistent.fq")
    assert r.returncode == 1 and "does not exist" not in r.stderr
    r = run("search", "--index", "x", "--fastx", "y", "--threads", "four")
    assert r.returncode == 1 and "failed to parse" in r.stderr


def test_commands_fail_loudly_without_a_gpu(tmp_path):
    """No silent CPU fallback in the binary: on a machine without a GPU `index` and `smooth` stop with a message
    unless the developer switch for their host code is given (tests/conftest.py gives it to the host-logic tests of
    this suite); `search` and `call` have no host path at all."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("SVDSS_INDEX_CPU", "SVDSS_SMOOTH_HOST")}
    fa = tmp_path / "r.fa"
    fa.write_text(">c\n" + "ACGT" * 500 + "\n")
    r = run("index", "-d", str(fa), "-o", str(tmp_path / "r.fmd"), env=env)
    assert r.returncode == 1 and "no GPU found" in r.stderr and not (tmp_path / "r.fmd").exists()
    bam = tmp_path / "x.bam"
    bam.write_bytes(bam_writer.bam([("c", 2000)], [bam_writer.record("q", 0, 0, 10, 60, [("M", 100)], "ACGT" * 25)]))
    r = subprocess.run([BIN, "smooth", "--reference", str(fa), "--bam", str(bam)], capture_output=True, timeout=120, env=env)
    assert r.returncode == 1 and b"no GPU found" in r.stderr and r.stdout == b""
    r = run("index", "-d", str(fa), "-o", str(tmp_path / "r.fmd"), env=dict(env, SVDSS_INDEX_CPU="1"))
    assert r.returncode == 0 and (tmp_path / "r.fmd").exists()
    r = run("search", "--index", str(tmp_path / "r.fmd"), "--fastx", str(fa), env=env)
    assert r.returncode == 1 and r.stdout == ""
    # the library's entry point as well (svdss_index_build: the Python mirror FMDIndex.build)
    import sys
    r = subprocess.run([sys.executable, "-c", "import numpy as np, svdss_amd\n"
                        "try:\n    svdss_amd.FMDIndex.build([np.ones(100, np.uint8)])\n    print('built')\n"
                        "except svdss_amd.SvdssError as e:\n    print('refused', e)"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert "refused" in r.stdout and "no GPU found" in r.stdout, r.stdout + r.stderr


def test_index_subcommand_roundtrip(tmp_path):
    ref = synth.make_reference([30000, 8000], seed=5, n_runs=(25,))
    fa = tmp_path / "ref.fa.gz"
    with gzip.open(fa, "wt") as fh:
        for i, c in enumerate(ref):
            s = synth.to_ascii(c)
            fh.write(f">chr{i} some description\n")
            for k in range(0, len(s), 70):
                fh.write(s[k:k + 70].lower() if i else s[k:k + 70])
                fh.write("\n")
    out = tmp_path / "ref.fa.fmd"
    r = run("index", "-t", "2", "-d", str(fa), "-o", str(out))             # run_svdss:142
    assert r.returncode == 0, r.stderr
    ix = svdss_amd.FMDIndex.load(str(out))
    jx = svdss_amd.FMDIndex.build(ref, threads=2)
    assert ix.size == jx.size and (ix.acc == jx.acc).all()
    w = ref[1][100:160]
    assert ix.count(w) == jx.count(w) >= 1


def _expected_text(names, reads, hps, searched, fm, assemble, T, bsize):
    """ping_pong.cpp:213-236 order: batches of bsize -> thread t = n % T -> std::map name order."""
    lines = []
    n = len(names)
    for b0 in range(0, n, bsize):
        for t in range(T):
            group = {}
            for k in range(b0 + t, min(n, b0 + bsize), T):
                if searched[k]:
                    group.setdefault(names[k], []).append(k)
            for name in sorted(group):
                first = True
                for k in group[name]:
                    raw, _ = fm.ping_pong_search(reads[k])
                    sfs = O.assemble(raw) if assemble else raw
                    for q, l in sfs:
                        lines.append(f"{name if first else '*'}\t{q}\t{l}\t{hps[k]}\t\n")
                        first = False
    return "".join(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("assemble", [True, False])
def test_search_fastx_and_bam_text(tmp_path, assemble):
    ref, hap, svs, flat, offs = small_workload(seed=81, n_reads=60, read_len=900, ref_lens=(90000,))
    fm = O.OracleFMD.build(ref)
    ix = svdss_amd.FMDIndex.build(ref)
    fmd = tmp_path / "ref.fmd"
    ix.save(str(fmd))
    reads = [flat[offs[i]:offs[i + 1]] for i in range(60)]
    reads[7] = reads[7].copy(); reads[7][40] = 5
    names = [f"m64/{(i * 37) % 60}/ccs" for i in range(60)]
    # ---- FASTQ (multi-line sequence, lower case, '@' in quality)
    fq = tmp_path / "reads.fq"
    with open(fq, "w") as fh:
        for nm, r in zip(names, reads):
            s = synth.to_ascii(r)
            fh.write(f"@{nm} extra\n{s[:300]}\n{s[300:].lower()}\n+\n{'@' * 300}\n{'I' * (len(s) - 300)}\n")
    extra = [] if assemble else ["--noassemble"]
    r = run("search", "--index", str(fmd), "--fastx", str(fq), "--threads", "3", "--bsize", "20", *extra)
    assert r.returncode == 0, r.stderr
    exp = _expected_text(names, reads, [0] * 60, [True] * 60, fm, assemble, 3, 18)   # bsize rounded to 18
    assert r.stdout == exp and len(exp) > 0
    # ---- BAM: XF/HP tags, filtered flags, short read
    recs, keep_names, keep_reads, hps, searched = [], [], [], [], []
    for i, (nm, rd) in enumerate(zip(names, reads)):
        flag = 0
        if i % 11 == 3:
            flag = 256      # secondary: dropped (ping_pong.cpp:66-69)
        if i % 13 == 5:
            flag = 2048     # supplementary: dropped
        if i == 20:
            flag = 4        # unmapped: dropped
        xf = 1 if i % 5 == 0 else 0
        hp = i % 3
        tags = [("NM", "i", 3)]
        if i % 2:
            tags.append(("XF", "C", xf))
        elif xf:
            tags.append(("XF", "i", xf))
        if hp:
            tags.append(("HP", "s", hp))
        tags.append(("RG", "Z", "grp"))
        recs.append(bam_writer.record(nm, flag, 0, 100 + i, 60, [("M", len(rd))], synth.to_ascii(rd), tags))
        if flag == 0:
            keep_names.append(nm); keep_reads.append(rd); hps.append(hp); searched.append(xf == 0)
    recs.insert(4, bam_writer.record("short", 0, 0, 50, 60, [("M", 50)], "ACGT" * 12 + "AC"))  # l_qseq < 100: dropped
    bam = tmp_path / "reads.bam"
    bam.write_bytes(bam_writer.bam([("chr1", 90000)], recs))
    r = run("search", "--index", str(fmd), "--bam", str(bam), "--threads", "4", "--bsize", "16", *extra)
    assert r.returncode == 0, r.stderr
    assert r.stdout == _expected_text(keep_names, keep_reads, hps, searched, fm, assemble, 4, 16)
    r2 = run("search", "--index", str(fmd), "--bam", str(bam), "--noputative", "--threads", "4", "--bsize", "16", *extra)
    assert r2.returncode == 0
    assert r2.stdout == _expected_text(keep_names, keep_reads, hps, [True] * len(keep_names), fm, assemble, 4, 16)
    assert len(r2.stdout) > len(r.stdout)
    # --gpus N (index replicated, batches go to whichever GPU is free, output in input order): the same bytes.  On a
    # one-GPU box SVDSS_GPUS_OVERSUBSCRIBE puts the replicas on the same device -- the code path is the same
    r3 = run("search", "--index", str(fmd), "--bam", str(bam), "--noputative", "--threads", "4", "--bsize", "16", "--gpus", "3",
             *extra, env=dict(os.environ, SVDSS_GPUS_OVERSUBSCRIBE="1"))
    assert r3.returncode == 0, r3.stderr
    assert r3.stdout == r2.stdout and "replicated on 3 GPUs" in r3.stderr
    # the text parses back (sfs.cpp:5-30)
    parsed = svdss_amd.parse_sfsfile(r2.stdout)
    assert set(parsed) <= set(keep_names)


def test_index_writes_an_rld0_fmd_and_the_records_beside_it(tmp_path):
    """`SVDSS index -d ref.fa -o ref.fa.fmd` (run_svdss:142): the .fmd is ropebwt3's rld0 dump (what upstream restores),
    `<fmd>.svdss` the records themselves (the index is rebuilt from them where it is made resident; SVDSS_INDEX_FULL=1:
    the full layout); all three restore to an index with the same BWT.  (Host builder: no GPU needed.)"""
    import ctypes as C
    from svdss_amd._lib import lib
    ref = synth.make_reference([20000, 3000], seed=9, n_runs=(50,))
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for i, c in enumerate(ref):
            s = synth.to_ascii(c)
            fh.write(f">chr{i + 1} x\n" + "\n".join(s[k:k + 70] for k in range(0, len(s), 70)) + "\n")
    fmd = tmp_path / "ref.fa.fmd"
    r = run("index", "-t", "2", "-d", str(fa), "-o", str(fmd))
    assert r.returncode == 0, r.stderr
    assert fmd.read_bytes()[:4] == b"RLD\x03"
    side = (tmp_path / "ref.fa.fmd.svdss").read_bytes()
    assert side[:8] == b"SVDSSRC1"
    # ~1 byte per base, not 19 -- and, since round 6, the rank blocks behind the records (64 bytes per 128 BWT symbols:
    # another byte per base): the index as a rank structure alone, for a `search` with few reads to search
    n_rec = 88 + 8 * 2 + 23000          # header, two record lengths, the records
    assert side[n_rec:n_rec + 8] == b"SVDSSBK1" and len(side) < 2 * 24000 + 1000
    r = run("index", "-t", "2", "-d", str(fa), "-o", str(tmp_path / "full.fmd"), env=dict(os.environ, SVDSS_INDEX_FULL="1"))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "full.fmd.svdss").read_bytes()[:8] == b"SVDSSFM2"
    want = svdss_amd.FMDIndex.build(ref, threads=2)
    n = C.c_int64()
    assert lib.svdss_fmd_read_bwt(str(fmd).encode(), None, 0, C.byref(n)) == 0 and n.value == want.size
    got = np.zeros(n.value, np.uint8)
    assert lib.svdss_fmd_read_bwt(str(fmd).encode(), got.ctypes.data, n.value, C.byref(n)) == 0
    assert (got == want.bwt()).all()
    lazy = svdss_amd.FMDIndex.load(str(fmd))                                       # through <fmd>.svdss
    assert lazy.size == want.size and (lazy.acc == want.acc).all()                 # known before anything is built
    assert (lazy.bwt() == want.bwt()).all() and lazy.count(ref[0][50:90]) == want.count(ref[0][50:90])
    assert (svdss_amd.FMDIndex.load(str(tmp_path / "full.fmd")).bwt() == want.bwt()).all()
    lazy.save_records(str(tmp_path / "again.rc"))
    assert (tmp_path / "again.rc").read_bytes() == side[:n_rec]
    want.save_records(str(tmp_path / "fromtext.rc"))                               # records out of the text
    assert (tmp_path / "fromtext.rc").read_bytes() == side[:n_rec]
    # the blocks section alone restores the rank structure: same BWT
    from svdss_amd._lib import check
    h = C.c_void_p()
    check(lib.svdss_index_load(str(fmd).encode(), C.byref(h)), "load")
    check(lib.svdss_index_attach_blocks(h, str(fmd).encode()), "attach")
    got2 = np.zeros(want.size, np.uint8)
    check(lib.svdss_index_bwt(h, got2.ctypes.data), "bwt")
    assert (got2 == want.bwt()).all()
    lib.svdss_index_free(h)
    h = C.c_void_p()
    check(lib.svdss_index_load(str(tmp_path / "full.fmd").encode(), C.byref(h)), "load")
    assert lib.svdss_index_attach_blocks(h, str(tmp_path / "full.fmd").encode()) == 1      # SVDSS_EINVAL: a full layout has no such section
    lib.svdss_index_free(h)
    os.remove(tmp_path / "ref.fa.fmd.svdss")
    back = svdss_amd.FMDIndex.load(str(fmd))                                       # through the rld0 import
    assert back.size == want.size and (back.acc == want.acc).all()


@pytest.mark.gpu
def test_search_bam_inflated_on_the_gpu_or_the_host_same_bytes(tmp_path):
    """The host path of `search --bam` (SVDSS_BAM_DEVICE=0: records sliced on the host; the default since round 4 handles
    them on the GPU, tests/test_bam_device_gpu.py): BGZF blocks inflated by csrc/inflate.hip, by the host workers, or
    half / half: the same text, and the text of the device path; a block whose content does not match its CRC32 footer
    ends the run with exit code 1 on every path."""
    import struct
    import zlib
    ref, hap, svs, flat, offs = small_workload(seed=83, n_reads=400, read_len=3000, ref_lens=(120000,))
    ix = svdss_amd.FMDIndex.build(ref)
    fmd = tmp_path / "ref.fmd"
    ix.save(str(fmd))
    rng = np.random.default_rng(4)
    recs = []
    for i in range(400):
        rd = flat[offs[i]:offs[i + 1]]
        qual = bytes(rng.integers(20, 60, size=len(rd), dtype=np.uint8).tolist())
        recs.append(bam_writer.record(f"r{i:04d}", 0, 0, 100 + i, 60, [("M", len(rd))], synth.to_ascii(rd), [("HP", "C", i % 3)], qual=qual))
    bam = tmp_path / "reads.bam"
    data = bam_writer.bam([("chr1", 120000)], recs)
    bam.write_bytes(data)
    assert len(data) > 300000          # several BGZF blocks, records straddling them
    outs = {}
    for mode in ("101", "0", "50", "100"):     # every chunk on the GPU / none / every other one / the default
        r = run("search", "--index", str(fmd), "--bam", str(bam), "--noputative", "--threads", "4", "--bsize", "64", "--verbose",
                env=dict(os.environ, SVDSS_GPU_INFLATE=mode, SVDSS_DEBUG="1", SVDSS_BAM_DEVICE="0"))
        assert r.returncode == 0, r.stderr
        outs[mode] = r.stdout
        if mode in ("101", "0"):
            assert ("inflated on the GPU" in r.stderr) == (mode == "101"), r.stderr
    assert outs["101"] == outs["0"] == outs["50"] == outs["100"] and outs["0"].count("\n") > 400
    r = run("search", "--index", str(fmd), "--bam", str(bam), "--noputative", "--threads", "4", "--bsize", "64", "--verbose")
    assert r.returncode == 0 and "device path" in r.stderr and r.stdout == outs["0"]
    # chunk buffers page-locked (forced: the file is far below the size at which the reader pins them), with no
    # page-locked memory to be had at all (cap 0: every buffer falls back to ordinary memory), several small chunks in
    # flight, and the k-mer table order left to the binary or fixed: the same text
    for env in ({"SVDSS_PIN_MIN_CHUNKS": "0"}, {"SVDSS_PIN_MIN_CHUNKS": "0", "SVDSS_PIN_CAP_GB": "0"},
                {"SVDSS_PIN_MIN_CHUNKS": "0", "SVDSS_BAM_SLAB_KB": "64"}, {"SVDSS_KMER": "16"}, {"SVDSS_KMER": "9"}):
        for dev in ("0", "1"):
            r = run("search", "--index", str(fmd), "--bam", str(bam), "--noputative", "--threads", "4", "--bsize", "64",
                    env=dict(os.environ, SVDSS_BAM_DEVICE=dev, **env))
            assert r.returncode == 0, (env, r.stderr[-300:])
            assert r.stdout == outs["0"], env
    # flip one bit in the middle of the second block's deflate stream
    bad = bytearray(data)
    first = struct.unpack_from("<H", data, 16)[0] + 1
    second = struct.unpack_from("<H", data, first + 16)[0] + 1
    bad[first + 18 + (second - 26) // 2] ^= 0x10
    (tmp_path / "bad.bam").write_bytes(bytes(bad))
    for mode in ("101", "0", "device"):
        env = dict(os.environ, SVDSS_GPU_INFLATE=mode, SVDSS_BAM_DEVICE="0") if mode != "device" else dict(os.environ)
        r = run("search", "--index", str(fmd), "--bam", str(tmp_path / "bad.bam"), "--noputative", env=env)
        assert r.returncode == 1 and ("CRC" in r.stderr or "inflate" in r.stderr), (mode, r.stderr[-300:])


@pytest.mark.gpu
def test_search_with_the_rank_blocks_alone_writes_the_same_text(tmp_path):
    """Round 6: a `search` that expects few reads to search makes the index resident as a rank structure alone
    (svdss_index_attach_blocks: the blocks `SVDSS index` leaves behind the records; no text, suffix array or k-mer table).
    Forced here both ways (SVDSS_SEARCH_LF=1 / 0) and left to the binary on a BAM in which one read in ten is to be
    searched: the same text as the full index gives, putative and not; an index without the section (SVDSS_INDEX_NO_BLOCKS)
    restores as before."""
    ref, hap, svs, flat, offs = small_workload(seed=85, n_reads=600, read_len=3000, ref_lens=(150000, 40000))
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as fh:
        for i, c in enumerate(ref):
            fh.write(f">chr{i + 1}\n{synth.to_ascii(c)}\n")
    fmd, fmd0 = tmp_path / "ref.fmd", tmp_path / "ref0.fmd"
    assert run("index", "-d", str(fa), "-o", str(fmd)).returncode == 0
    assert run("index", "-d", str(fa), "-o", str(fmd0), env=dict(os.environ, SVDSS_INDEX_NO_BLOCKS="1")).returncode == 0
    assert os.path.getsize(str(fmd0) + ".svdss") < os.path.getsize(str(fmd) + ".svdss")
    rng = np.random.default_rng(5)
    recs = []
    for i in range(600):
        rd = flat[offs[i]:offs[i + 1]]
        recs.append(bam_writer.record(f"r{i:04d}", 0, 0, 100 + i, 60, [("M", len(rd))], synth.to_ascii(rd), [("XF", "C", 0 if i % 10 == 0 else 1)]))
    bam = tmp_path / "reads.bam"
    bam.write_bytes(bam_writer.bam([("chr1", 150000), ("chr2", 40000)], recs))
    for extra in ((), ("--noputative",)):
        base = run("search", "--index", str(fmd), "--bam", str(bam), "--threads", "4", "--bsize", "64", "--verbose", *extra,
                   env=dict(os.environ, SVDSS_SEARCH_LF="0", SVDSS_SEARCH_EARLY="1"))
        assert base.returncode == 0 and "rank blocks alone" not in base.stderr and base.stdout.count("\n") > 50
        for env in ({"SVDSS_SEARCH_LF": "1"}, {"SVDSS_SEARCH_LF": "1", "SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_BATCH_MB": "1", "SVDSS_PARK_GROUP_READS": "50"},
                    {"SVDSS_SEARCH_LF": "1", "SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_BATCH_MB": "1", "SVDSS_PARK_MB": "1", "SVDSS_PARK_ARENA_MB": "1"}, {}):
            r = run("search", "--index", str(fmd), "--bam", str(bam), "--threads", "4", "--bsize", "64", "--verbose", *extra, env=dict(os.environ, SVDSS_SEARCH_EARLY="1", **env))
            assert r.returncode == 0, r.stderr[-600:]
            assert r.stdout == base.stdout, (extra, env)
            if env:
                assert "rank blocks alone" in r.stderr, r.stderr[-800:]
        r0 = run("search", "--index", str(fmd0), "--bam", str(bam), "--threads", "4", "--bsize", "64", "--verbose", *extra,
                 env=dict(os.environ, SVDSS_SEARCH_LF="1", SVDSS_SEARCH_EARLY="1"))
        assert r0.returncode == 0 and "rank blocks alone" not in r0.stderr and r0.stdout == base.stdout


def test_fastx_reader_line_shapes(tmp_path, monkeypatch):
    """csrc/fastx_reader.h (kseq's role, chromosomes.cpp:9-27, fastq.hpp:17-35) reads the file in blocks and finds lines
    with memchr: 60-column lines, one base per line, a chromosome on one line, CRLF, blank lines and stray text in front
    of the first header, gzip, lower case, FASTQ with quality lines that start with '@' and span lines -- `SVDSS index`
    of each of them is the index of the same records."""
    monkeypatch.setenv("SVDSS_INDEX_CPU", "1")
    ref = synth.make_reference([30011, 7, 1, 8000], seed=5, n_runs=(25,))
    want = svdss_amd.FMDIndex.build(ref, threads=4).bwt()

    def write(path, width, crlf, blank, gz, lower, fastq):
        nl = "\r\n" if crlf else "\n"
        with (gzip.open if gz else open)(path, "wt", newline="") as fh:
            if blank:
                fh.write(nl + "junk line" + nl)
            for i, c in enumerate(ref):
                s = synth.to_ascii(c)
                if lower and i % 2:
                    s = s.lower()
                fh.write((f"@chr{i} d" if fastq else f">chr{i}\tdesc") + nl)
                for k in range(0, len(s), width):
                    fh.write(s[k:k + width] + nl)
                if fastq:
                    fh.write("+" + nl)
                    for k in range(0, len(s), width):
                        fh.write("@" * len(s[k:k + width]) + nl)
                elif blank and i == 0:
                    fh.write(nl)

    shapes = [(60, False, False, False, False, False), (70, True, True, True, True, False), (10 ** 9, False, False, False, True, False),
              (1, False, False, False, False, False), (61, True, False, False, False, True), (4096, False, True, True, False, True)]
    for n, (width, crlf, blank, gz, lower, fastq) in enumerate(shapes):
        p = tmp_path / (f"f{n}.fa" + (".gz" if gz else ""))
        write(p, width, crlf, blank, gz, lower, fastq)
        out = tmp_path / f"f{n}.fmd"
        r = run("index", "-t", "4", "-d", str(p), "-o", str(out))
        assert r.returncode == 0, r.stderr
        assert (svdss_amd.FMDIndex.load(str(out)).bwt() == want).all(), shapes[n]
