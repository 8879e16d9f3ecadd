"""svdss_bam_batch_run (csrc/bam_device.hip): compressed BGZF blocks in, names / tags / SFS out, with the record chain,
the filters of ping_pong.cpp:66-79, the XF / HP lookup of :196-203 and the 4-bit -> nt6 expansion of :90-94 on the GPU.
Checked against the records as the test wrote them and the oracle's search of the same reads; batch sizes and segment
sizes are forced small so that records straddle blocks, segments and batches, and the segment guesses go wrong."""
import os
import struct
import zlib

import numpy as np
import pytest

import svdss_amd
from svdss_amd import bamdev, bgzf, synth
from svdss_amd._lib import SvdssError
from tests import bam_writer, oracle_lib as O
from tests.common import small_workload

pytestmark = pytest.mark.gpu


def _records(names, reads, rng, decoys=False):
    """records with every filter / tag shape; returns (record bytes list, expected slots [(name, hp, searched, read)])"""
    recs, slots = [], []
    for i, (nm, rd) in enumerate(zip(names, reads)):
        flag = 0
        if i % 11 == 3:
            flag = 256
        if i % 13 == 5:
            flag = 2048
        if i % 17 == 9:
            flag = 4
        if i % 19 == 2:
            flag = 16          # reverse strand: kept
        xf = 1 if i % 5 == 0 else 0
        hp = i % 3
        tags = [("NM", "i", 3)]
        if i % 2:
            tags.append(("XF", "C", xf))
        elif xf:
            tags.append(("XF", "i", xf))
        if i % 7 == 1:
            tags.append(("MD", "Z", "12A" * (i % 40)))
        if hp:
            tags.append(("HP", ["s", "c", "I"][i % 3], hp))
        tags.append(("RG", "Z", "grp"))
        qual = None
        if decoys and i % 4 == 0:
            # qualities that look like a chain of records (block_size, refID, pos, l_read_name ...): the segment guesser
            # may take them for one; the link step must not
            def mini(l_seq, name):
                body = struct.pack("<iiBBHHHiiii", 0, 5, len(name), 60, 4680, 0, 0, l_seq, -1, -1, 0) + name + b"\x11" * ((l_seq + 1) // 2) + b"\x20" * l_seq
                return struct.pack("<i", len(body)) + body
            fake = mini(8, b"abc\0") + mini(0, b"de\0") + mini(2, b"f\0")
            q = bytearray(rng.integers(20, 60, size=len(rd), dtype=np.uint8).tobytes())
            for at in range(10, len(q) - len(fake), 700):
                q[at:at + len(fake)] = fake
            qual = bytes(q)
        recs.append(bam_writer.record(nm, flag, 0, 100 + i, 60, [("M", len(rd))], synth.to_ascii(rd), tags, qual=qual))
        if flag in (0, 16) and len(rd) >= 100:
            slots.append((nm, hp, xf == 0, rd))
    return recs, slots


def _bgzf_levels(data, rng, block=60000):
    """BGZF members of varying size and zlib level (stored blocks included)"""
    out, i = [], 0
    while i < len(data):
        n = int(rng.integers(1, block))
        piece = data[i:i + n]
        i += n
        comp = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15)
        c = comp.compress(piece) + comp.flush()
        out.append(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(c) + 25) + c +
                   struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
        if rng.random() < 0.05:
            out.append(bam_writer._bgzf_block(b""))      # empty members in the middle of the file are legal
    out.append(bam_writer._bgzf_block(b""))
    return b"".join(out)


def _raw_bam(refs, recs):
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    return hdr + b"".join(recs)


@pytest.fixture(scope="module")
def case():
    ref, hap, svs, flat, offs = small_workload(seed=91, n_reads=400, read_len=1500, ref_lens=(150000,))
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    fm = O.OracleFMD.build(ref)
    reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(400)]
    reads[7][40] = 5
    reads[11] = reads[11][:99]          # l_qseq < 100: dropped, counted
    reads[12] = reads[12][:100]         # kept
    reads[13] = reads[13][:101]         # odd length: the last nibble
    reads[14] = np.concatenate([reads[14], reads[15], reads[16], reads[17]])   # a record larger than a BGZF block's share
    names = [f"m64/{(i * 37) % 400}/ccs" for i in range(400)]
    names[21] = "x"
    return ref, ix, fm, reads, names


def _check(out, slots, fm, assemble, putative):
    assert [(o[0], o[1]) for o in out] == [(s[0], s[1]) for s in slots]
    for (nm, hp, sfs), (_, _, srch, rd) in zip(out, slots):
        if putative and not srch:
            assert sfs is None
            continue
        raw, _ = fm.ping_pong_search(rd)
        exp = O.assemble(raw) if assemble else raw
        assert sfs == [tuple(x) for x in exp], nm


@pytest.mark.parametrize("assemble,putative", [(True, True), (False, False)])
def test_whole_file_in_one_batch(case, assemble, putative):
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(5)
    recs, slots = _records(names, reads, rng)
    data = bam_writer.bam([("chr1", 150000)], recs)
    out, st = bamdev.search_bam(ix, data, assemble=assemble, putative=putative)
    assert st["records"] == len(recs) and st["short"] == 1
    _check(out, slots, fm, assemble, putative)


@pytest.mark.parametrize("batch_kb,seg_kb", [(64, 1), (200, 4), (1000, 16), (5, 1)])
def test_records_straddle_blocks_segments_and_batches(case, batch_kb, seg_kb, monkeypatch):
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(batch_kb)
    recs, slots = _records(names, reads, rng, decoys=True)
    data = _bgzf_levels(_raw_bam([("chr1", 150000)], recs), rng, block=20000)
    monkeypatch.setenv("SVDSS_BAM_SEG_KB", str(seg_kb))
    out, st = bamdev.search_bam(ix, data, assemble=True, putative=True, batch_bytes=batch_kb << 10)
    assert st["records"] == len(recs) and st["short"] == 1 and (st["batches"] > 3 or batch_kb >= 1000)
    _check(out, slots, fm, True, True)
    # most guesses hold; the decoys and the long record make some fail, and the result does not care
    assert st["segments"] > st["batches"] and st["rewalked"] < st["segments"]


def test_header_only_and_empty_inputs(case):
    ref, ix, fm, reads, names = case
    out, st = bamdev.search_bam(ix, bam_writer.bam([("chr1", 150000)], []))
    assert out == [] and st["records"] == 0
    rec = bam_writer.record("only", 0, 0, 5, 60, [("M", len(reads[0]))], synth.to_ascii(reads[0]))
    out, st = bamdev.search_bam(ix, bam_writer.bam([("chr1", 150000), ("chr2", 5)], [rec]), batch_bytes=1 << 10)
    raw, _ = fm.ping_pong_search(reads[0])
    assert [(o[0], o[1], o[2]) for o in out] == [("only", 0, [tuple(x) for x in O.assemble(raw)])]


def test_damage_is_reported(case):
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(3)
    recs, _ = _records(names[:40], reads[:40], rng)
    raw = _raw_bam([("chr1", 150000)], recs)
    good = bam_writer.bgzf(raw, block=30000)
    # a flipped bit inside a deflate stream: inflate or CRC must say so
    blocks = bgzf.bgzf_blocks(good)
    bad = bytearray(good)
    coff, clen, isize, crc = blocks[len(blocks) // 2]
    bad[coff + clen // 2] ^= 0x10
    with pytest.raises(SvdssError) as e:
        bamdev.search_bam(ix, bytes(bad))
    assert e.value.detail in ("BGZF inflate failed", "BGZF block CRC mismatch")
    # a wrong CRC in a footer
    coff, clen, isize, crc = blocks[1]
    bad = bytearray(good)
    bad[coff + clen:coff + clen + 4] = struct.pack("<I", crc ^ 1)
    with pytest.raises(SvdssError) as e:
        bamdev.search_bam(ix, bytes(bad))
    assert e.value.detail == "BGZF block CRC mismatch"
    # the file ends inside a record
    cut = bam_writer.bgzf(raw[:-37], block=30000)
    with pytest.raises(SvdssError) as e:
        bamdev.search_bam(ix, cut, batch_bytes=20 << 10)
    assert e.value.detail == "truncated record"
    # a mapped read without a reference (ping_pong.cpp:76-79)
    recs2 = list(recs)
    recs2[5] = bam_writer.record("notid", 0, -1, 5, 60, [("M", len(reads[5]))], synth.to_ascii(reads[5]))
    with pytest.raises(SvdssError) as e:
        bamdev.search_bam(ix, bam_writer.bam([("chr1", 150000)], recs2))
    assert "core.tid < 0" in e.value.detail
    # l_seq that does not fit its record
    r6 = bytearray(recs[6])
    r6[4 + 16:4 + 20] = struct.pack("<i", 10_000_000)
    recs3 = list(recs)
    recs3[6] = bytes(r6)
    with pytest.raises(SvdssError) as e:
        bamdev.search_bam(ix, bam_writer.bam([("chr1", 150000)], recs3))
    assert e.value.detail == "corrupt record"


def test_crc32_kernel_on_every_block_size(case):
    """the footer check alone: members of 1 .. 65536 bytes, unaligned starts (a CRC mismatch would raise)"""
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(11)
    body = _raw_bam([("chr1", 150000)], [])
    filler = bytes(rng.integers(0, 4, size=70000, dtype=np.uint8) * 67 + 1)   # (compressible: a member of 65,536 bytes must fit 64 KB)
    # one big unmapped record per size class so that the chain stays valid whatever the member sizes are
    recs = []
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 4095, 4097, 65535 - 40, 65536 - 40):
        q = filler[:n]
        recs.append(bam_writer.record("f%d" % n, 4, -1, -1, 0, [], "", qual=None) + b"")
        core = struct.pack("<iiBBHHHiiii", -1, -1, 2, 0, 4680, 0, 4, 0, -1, -1, 0) + b"f\0" + b"ZZZ" + q + b"\0"
        recs.append(struct.pack("<i", len(core)) + core)
    data = body + b"".join(recs)
    out = []
    i = 0
    sizes = [1, 2, 3, 4, 5, 7, 63, 64, 65, 100, 255, 256, 257, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 30000, 65535, 65536]
    k = 0
    while i < len(data):
        n = sizes[k % len(sizes)]
        k += 1
        out.append(bam_writer._bgzf_block(data[i:i + n]))
        i += n
    out.append(bam_writer._bgzf_block(b""))
    res, st = bamdev.search_bam(ix, b"".join(out))
    assert res == [] and st["records"] == len(recs)


def test_binary_device_path_writes_the_host_paths_bytes(case, tmp_path):
    """`SVDSS search --bam`: records handled on the GPU (default) against SVDSS_BAM_DEVICE=0 (records sliced on the
    host) -- the same stdout byte for byte, with slabs / batches so small that the file is dozens of batches, with
    several feeding threads per GPU and two (oversubscribed) GPUs, for two --threads / --bsize settings (the text order
    depends on them, ping_pong.cpp:213-236: device batches never end where reference batches do)."""
    import re
    import subprocess
    from tests.common import BIN
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(23)
    # 3,200 records: the 400 reads eight times under other names
    recs = []
    for rep in range(8):
        r, _ = _records([f"{n}/{rep}" for n in names], reads, rng, decoys=(rep % 2 == 0))
        recs += r
    bam = tmp_path / "reads.bam"
    bam.write_bytes(_bgzf_levels(_raw_bam([("chr1", 150000)], recs), rng, block=60000))
    fmd = tmp_path / "ref.fmd"
    ix.save(str(fmd))

    def run(env, *extra):
        r = subprocess.run([BIN, "search", "--index", str(fmd), "--bam", str(bam), "--verbose", *extra], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        return r
    for extra in (("--threads", "4", "--bsize", "100"), ("--threads", "3", "--bsize", "1000", "--noputative", "--noassemble")):
        host = run({"SVDSS_BAM_DEVICE": "0"}, *extra)
        assert "device path" not in host.stderr and host.stdout.count("\n") > 5000
        dev = run({"SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SEG_KB": "8"}, *extra)
        assert "device path" in dev.stderr and dev.stdout == host.stdout
        assert dev.stderr.count("Alignment filtered due to l_qseq") == host.stderr.count("Alignment filtered due to l_qseq") == 8
        dev2 = run({"SVDSS_BAM_SLAB_KB": "128", "SVDSS_BAM_BATCH_MB": "2", "SVDSS_GPUS_OVERSUBSCRIBE": "1", "SVDSS_SEARCH_FEEDERS": "3"},
                   "--gpus", "2", *extra)
        assert dev2.stdout == host.stdout
        big = run({}, *extra)          # the default sizes: one batch
        assert big.stdout == host.stdout
        # round 6: the front end runs beside the index restore (one GPU: the default) -- the batches read before the index
        # is resident park their unpacked reads in HBM and are searched one large launch per group.  The index of this
        # test is resident in milliseconds, so it is held back (SVDSS_EARLY_HOLD_MS): dozens of batches get parked, in
        # groups of a few hundred reads, in arenas of 1 MB (several), with a park too small for all of them (the rest
        # waits for the index), and against the index-first order (SVDSS_SEARCH_EARLY=0)
        first = run({"SVDSS_SEARCH_EARLY": "0", "SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_BATCH_MB": "1"}, *extra)
        assert first.stdout == host.stdout and "front end beside the index restore" not in first.stderr
        for env in ({"SVDSS_PARK_GROUP_READS": "300", "SVDSS_PARK_ARENA_MB": "1"},
                    {"SVDSS_PARK_GROUP_READS": "100000"},
                    {"SVDSS_PARK_MB": "2", "SVDSS_PARK_ARENA_MB": "1", "SVDSS_PARK_GROUP_READS": "200"},
                    {"SVDSS_PARK_GROUP_READS": "7", "SVDSS_SEARCH_FEEDERS": "2"}):
            ev = dict(env, SVDSS_SEARCH_EARLY="1", SVDSS_EARLY_HOLD_MS="1500", SVDSS_BAM_SLAB_KB="64", SVDSS_BAM_BATCH_MB="1")
            early = run(ev, *extra)
            m = re.search(r"front end beside the index restore: (\d+) batches \((\d+) records\) .* their (\d+) reads searched in (\d+) launch", early.stderr)
            assert m and int(m.group(1)) >= 2 and int(m.group(4)) >= 1, early.stderr[-1500:]
            if "SVDSS_PARK_MB" in env:
                assert int(m.group(3)) < 3000          # (the park was full: not every read fitted)
            assert early.stdout == host.stdout, env
            assert early.stderr.count("Alignment filtered due to l_qseq") == 8
        # round 5: --gpus N cuts the file into N regions, each with its own scanner / batcher / feeders; a region's first
        # record is guessed and proved at the seam.  SVDSS_REGION_TEST: 1 = every guess is no record (the regions fail
        # and run again from the region before), 2 = the seams do not fit (the regions run again)
        import re
        for n, knob in ((4, None), (3, "1"), (2, "2"), (7, None)):
            env = {"SVDSS_BAM_SLAB_KB": "64", "SVDSS_BAM_BATCH_MB": "1", "SVDSS_GPUS_OVERSUBSCRIBE": "1", "SVDSS_SEARCH_FEEDERS": "2",
                   "SVDSS_REGION_MIN_KB": "128"}
            if knob:
                env["SVDSS_REGION_TEST"] = knob
            sh = run(env, "--gpus", str(n), *extra)
            m = re.search(r"(\d+) regions of the file, one per GPU: (\d+) seam\(s\) run, (\d+) region\(s\) run again", sh.stderr)
            assert m and int(m.group(1)) == n, sh.stderr[-2000:]
            assert sh.stdout == host.stdout, (n, knob)
            assert sh.stderr.count("Alignment filtered due to l_qseq") == 8
            if knob is None:
                assert int(m.group(2)) >= n - 2 and m.group(3) == "0", [l for l in sh.stderr.splitlines() if "region" in l]
            elif knob == "1":
                assert m.group(3) == str(n - 1)
            else:
                assert int(m.group(3)) >= 1
    # damage through the binary: message + exit 1
    data = bytearray(bam.read_bytes())
    data[len(data) // 2] ^= 0x40
    (tmp_path / "bad.bam").write_bytes(bytes(data))
    r = subprocess.run([BIN, "search", "--index", str(fmd), "--bam", str(tmp_path / "bad.bam")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "critical" in r.stderr


def test_select_records_by_name_and_by_region(case):
    """svdss_bam_select_run, what `SVDSS call` keeps of a BAM: against the same filters applied in Python to the records
    as written (clusterer.cpp:118-122, :535-545), with batches so small that records straddle them."""
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(41)
    recs, metas = [], []
    for i, (nm, rd) in enumerate(zip(names, reads)):
        flag = [0, 16, 256, 2048, 4, 0, 0, 1024][i % 8]
        mapq = [60, 5, 20, 19, 0, 33][i % 6]
        pos = int(rng.integers(0, 140000))
        # CIGARs with every reference-consuming and query-consuming operation
        L = len(rd)
        if i % 3 == 0:
            cigar = [("S", 5), ("M", L - 25), ("D", 40), ("I", 10), ("M", 10)]
        elif i % 3 == 1:
            cigar = [("M", L // 2), ("N", 300), ("=", L - L // 2 - 3), ("X", 3)]
        else:
            cigar = [("M", L)]
        ref_len = sum(l for op, l in cigar if op in "MDN=X")
        recs.append(bam_writer.record(nm, flag, 0, pos, mapq, cigar, synth.to_ascii(rd), [("HP", "C", i % 3)]))
        metas.append((nm, flag, mapq, pos, pos + max(ref_len, 1)))
    data = _bgzf_levels(_raw_bam([("chr1", 150000)], recs), rng, block=30000)
    bodies = [r[4:] for r in recs]

    def expect(min_mapq, wanted=None, regions=None):
        out = []
        for body, (nm, flag, mapq, a, b) in zip(bodies, metas):
            if flag & (4 | 256 | 2048) or mapq < min_mapq:
                continue
            if wanted is not None or regions is not None:
                hit = wanted is not None and nm in wanted
                hit = hit or (regions is not None and any(t == 0 and a < e and b > s for t, s, e in regions))
                if not hit:
                    continue
            out.append(body)
        return out
    wanted = [names[i] for i in range(0, 400, 7)] + ["not/in/the/file"]
    got, st = bamdev.select_bam(data, names=wanted, min_mapq=20, batch_bytes=40 << 10)
    assert st["records"] == 400 and st["batches"] > 5
    assert got == expect(20, wanted=set(wanted))
    regions = sorted((0, int(s), int(s) + int(rng.integers(1, 3000))) for s in rng.integers(0, 150000, size=25))
    got, st = bamdev.select_bam(data, regions=regions, min_mapq=0, batch_bytes=100 << 10)
    assert got == expect(0, regions=regions) and 0 < len(got) < 300
    got, _ = bamdev.select_bam(data, names=wanted, regions=regions, min_mapq=20)
    assert got == expect(20, wanted=set(wanted), regions=regions)
    got, _ = bamdev.select_bam(data, min_mapq=30)           # no names, no regions: the flag / mapq filters alone
    assert got == expect(30)
    # a region that ends exactly where an alignment begins does not overlap it; one base more does
    nm0, f0, q0, a0, b0 = next(m for m in metas if not m[1] & (4 | 256 | 2048))
    g1, _ = bamdev.select_bam(data, regions=[(0, max(a0 - 10, 0), a0)])
    g2, _ = bamdev.select_bam(data, regions=[(0, max(a0 - 10, 0), a0 + 1)])
    assert g1 == expect(0, regions=[(0, max(a0 - 10, 0), a0)]) and g2 == expect(0, regions=[(0, max(a0 - 10, 0), a0 + 1)])
    assert len(g2) > len(g1)


def test_one_pass_for_call_the_store_keeps_slim_records_for_the_second_pass(case):
    """svdss_bam_select_store_run + svdss_bam_store_select (round 6): the first pass' records by name as before, and --
    from records kept in HBM -- the second pass' records by region, slim: core | name | CIGAR | bases | HP as an int32 tag
    when the record had an integer one (every integer type, a string-typed HP, none), other tags and qualities dropped;
    batches and store arenas so small that there are dozens; a store too small stays incomplete."""
    ref, ix, fm, reads, names = case
    rng = np.random.default_rng(43)
    recs, metas, slims = [], [], []
    for i, (nm, rd) in enumerate(zip(names, reads)):
        flag = [0, 16, 256, 2048, 4, 0, 0, 1024][i % 8]
        mapq = [60, 5, 20, 19, 0, 33][i % 6]
        pos = int(rng.integers(0, 140000))
        L = len(rd)
        cigar = [("S", 5), ("M", L - 25), ("D", 40), ("I", 10), ("M", 10)] if i % 3 == 0 else [("M", L // 2), ("N", 300), ("=", L - L // 2 - 3), ("X", 3)] if i % 3 == 1 else [("M", L)]
        ref_len = sum(l for op, l in cigar if op in "MDN=X")
        hp_kind = i % 9
        tags = [("NM", "i", 7), ("MD", "Z", "10A5")]
        hp = None
        if hp_kind < 6:
            ty = "cCsSiI"[hp_kind]
            hp = {"c": -3, "C": 200, "s": -300, "S": 40000, "i": -70000, "I": 3000000000}[ty]
            tags.insert(1, ("HP", ty, hp))
        elif hp_kind == 6:
            tags.append(("HP", "Z", "two"))                 # not an integer: bam_aux2i answers 0 = "no tag" for the caller
        qual = bytes(rng.integers(20, 60, size=L, dtype=np.uint8).tolist()) if i % 2 else None
        rec = bam_writer.record(nm, flag, 0, pos, mapq, cigar, synth.to_ascii(rd), tags, qual=qual)
        recs.append(rec)
        metas.append((nm, flag, mapq, pos, pos + max(ref_len, 1)))
        body = rec[4:]
        l_name, n_cig = body[8], struct.unpack_from("<H", body, 12)[0]
        head = 32 + l_name + 4 * n_cig + (L + 1) // 2
        tag = b"" if hp is None else b"HP" + (b"I" + struct.pack("<I", hp) if hp > 2147483647 else b"i" + struct.pack("<i", hp))
        slims.append(body[:head] + tag)
    data = _bgzf_levels(_raw_bam([("chr1", 150000)], recs), rng, block=30000)
    bodies = [r[4:] for r in recs]
    wanted = [names[i] for i in range(0, 400, 5)]
    regions = sorted((0, int(s), int(s) + int(rng.integers(1, 3000))) for s in rng.integers(0, 150000, size=25))
    for min_mapq in (0, 20):
        os.environ["SVDSS_STORE_ARENA_MB"] = "1"
        try:
            named, slim, st = bamdev.select_bam_store(data, wanted, regions, min_mapq=min_mapq, batch_bytes=40 << 10)
        finally:
            del os.environ["SVDSS_STORE_ARENA_MB"]
        ok = [not (m[1] & (4 | 256 | 2048)) and m[2] >= min_mapq for m in metas]
        assert st["complete"] == 1 and st["stored_batches"] == st["batches"] > 5 and st["stored_records"] == sum(ok)
        assert named == [sl for sl, m, k in zip(slims, metas, ok) if k and m[0] in set(wanted)] and all(st["named_slim"])
        want = [sl for sl, m, k in zip(slims, metas, ok) if k and any(m[3] < e and m[4] > s_ for _, s_, e in regions)]
        assert slim == want and 0 < len(want) < 300
    named, slim, st = bamdev.select_bam_store(data, wanted, regions, min_mapq=0, batch_bytes=40 << 10, max_store_bytes=1 << 16)
    assert st["complete"] == 0 and slim == []
    # (the batches behind the one that did not fit come whole)
    assert named == [(sl if f else b) for sl, b, m, f in zip([x for x, m in zip(slims, metas) if not (m[1] & (4 | 256 | 2048)) and m[0] in set(wanted)],
                                                            [x for x, m in zip(bodies, metas) if not (m[1] & (4 | 256 | 2048)) and m[0] in set(wanted)],
                                                            [m for m in metas if not (m[1] & (4 | 256 | 2048)) and m[0] in set(wanted)], st["named_slim"])]
    assert 0 in st["named_slim"]
