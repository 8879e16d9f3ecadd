"""An rld0 (`.fmd`) ENCODER and a multi-string BWT in ropebwt's convention, in Python, for tests: written from the
published format description (rld0.c of fermi / ropebwt2 / ropebwt3, version 3 -- [UPSTREAM-UNVERIFIED], no ropebwt3
here to run), independently of csrc/rld0.cpp.  Test infrastructure only.

`collection_bwt` orders the sentinels the way ropebwt does -- string i's '$' is smaller than string j's for i < j, so no
comparison ever runs past a sentinel -- which is NOT the order of this library's own index (sentinels ordered by the
text that follows): an `.fmd` produced here exercises the import path (decode -> recover the strings by LF walks ->
rebuild) on a file the library's own writer could not have written."""
import struct

import numpy as np


def collection_bwt(strings):
    """BWT (nt6 symbols, '$' = 0) of the collection s_0 $_0 s_1 $_1 ... with $_0 < $_1 < ... < A.  Suffix sorting by
    prefix doubling over the concatenation with distinct sentinel symbols (fine for ~10^5 symbols)."""
    m = len(strings)
    parts = []
    for i, s in enumerate(strings):
        parts.append(np.asarray(s, np.int64) + m)        # symbols above every sentinel
        parts.append(np.array([i], np.int64))            # $_i
    t = np.concatenate(parts)
    n = len(t)
    rank = t.copy()
    sa = np.argsort(rank, kind="stable")
    h = 1
    while True:
        r2 = np.full(n, -1, np.int64)
        r2[:n - h] = rank[h:]
        sa = np.lexsort((r2, rank))
        key_change = (rank[sa][1:] != rank[sa][:-1]) | (r2[sa][1:] != r2[sa][:-1])
        new = np.zeros(n, np.int64)
        new[sa] = np.concatenate([[0], np.cumsum(key_change)])
        rank = new
        if rank.max() == n - 1:
            break
        h *= 2
    prev = t[(sa - 1) % n]
    return np.where(prev < m, 0, prev - m).astype(np.uint8)


def _delta(x):
    y = x.bit_length() - 1
    z = (y + 1).bit_length() - 1
    return "0" * z + format(y + 1, f"0{z + 1}b") + (format(x ^ (1 << y), f"0{y}b") if y else "")


def encode_rld0(bwt, asize=6, sbits=3):
    """bytes of an rld0 file holding `bwt`.  Small blocks of 2^sbits words: counts of the previous block (16-bit when
    the block's total is below 0x4000, else 32-bit; type in the top two bits of word 0), then delta-coded run lengths
    + 3-bit symbols; a code never straddles two blocks; one closing header after the last run; rank frames."""
    bwt = np.asarray(bwt, np.uint8)
    ssize, abits = 1 << sbits, 3
    lsize = 1 << 23
    cut = np.flatnonzero(np.concatenate([[True], bwt[1:] != bwt[:-1]]))
    lens = np.diff(np.concatenate([cut, [len(bwt)]]))
    syms = bwt[cut]
    words = []                                             # the data words
    cnt = [0] * (asize + 1)
    mcnt = [0] * (asize + 1)
    blocks = []                                            # (header words, bit string of the block's codes)

    def tail_words(shead, hdr):
        stail = shead + ssize - (2 if (shead + ssize) % lsize == 0 else 1)
        return stail - (shead + hdr) + 1

    shead, hdr_words, bits = 0, [0, 0], ""
    cap = 64 * tail_words(0, 2)
    for l, c in zip(lens.tolist(), syms.tolist()):
        code = _delta(l) + format(c, f"0{abits}b")
        if len(bits) + len(code) > cap:
            blocks.append((hdr_words, bits))
            shead += ssize
            d = [cnt[i] - mcnt[i] for i in range(asize + 1)]
            if d[0] < 0x4000:
                raw = struct.pack(f"<{asize + 1}H", *d).ljust(16, b"\0")
                typ, nh = 0, 2
            else:
                raw = struct.pack(f"<{asize + 1}I", *d).ljust(32, b"\0")
                typ, nh = 1, 4
            hdr_words = list(struct.unpack(f"<{nh}Q", raw))
            hdr_words[0] |= typ << 62
            mcnt = list(cnt)
            bits = ""
            cap = 64 * tail_words(shead, nh)
        bits += code
        cnt[0] += l
        cnt[c + 1] += l
    blocks.append((hdr_words, bits))
    # the closing header
    d = [cnt[i] - mcnt[i] for i in range(asize + 1)]
    if d[0] < 0x4000:
        raw, typ, nh = struct.pack(f"<{asize + 1}H", *d).ljust(16, b"\0"), 0, 2
    else:
        raw, typ, nh = struct.pack(f"<{asize + 1}I", *d).ljust(32, b"\0"), 1, 4
    closing = list(struct.unpack(f"<{nh}Q", raw))
    closing[0] |= typ << 62
    for hdr, b in blocks:
        body = [int(b[i:i + 64].ljust(64, "0"), 2) for i in range(0, len(b), 64)]
        w = hdr + body
        assert len(w) <= ssize
        words += w + [0] * (ssize - len(w))
    words += closing
    k = len(words)
    # rank frames: one per 2^ibits positions -- (offset of the first block at or past it, symbol counts before it)
    total = cnt[0]
    n_blks = k // ssize + 1
    ibits = max(1, total // n_blks).bit_length() - 1 + 4
    n_frames = ((total + (1 << ibits) - 1) >> ibits) + 1
    frames = [[0] * (asize + 1) for _ in range(n_frames)]
    run = [0] * asize
    fk = 1
    last = (k >> sbits) << sbits
    for i in range(ssize, last + 1, ssize):
        w0 = words[i]
        if w0 >> 62:
            h = [x & 0x3fffffff for x in struct.unpack_from(f"<{asize + 1}I", struct.pack("<4Q", *words[i:i + 4]))]
        else:
            h = list(struct.unpack_from(f"<{asize + 1}H", struct.pack("<2Q", *(words + [0])[i:i + 2])))
        for j in range(asize):
            run[j] += h[j + 1]
        s = sum(run)
        while s >= (fk << ibits):
            fk += 1
        if fk < n_frames:
            frames[fk] = [i] + list(run)
    for f in range(1, n_frames):
        if frames[f][0] == 0:
            frames[f] = list(frames[f - 1])
    out = b"RLD\x03" + struct.pack("<IQQ", asize << 16 | sbits, k, n_frames)
    out += struct.pack(f"<{asize}Q", *cnt[1:])
    out += struct.pack(f"<{k}Q", *words)
    for fr in frames:
        out += struct.pack(f"<{asize + 1}Q", *fr)
    return out
