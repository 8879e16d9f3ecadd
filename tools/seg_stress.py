#!/usr/bin/env python3
"""Stress of the segmented search kernel: T threads, each with its own batch object, search the same small batch R times
with SVDSS_SEGMENTS segments per read (concurrent launches on one GPU); every result is compared with the one-lane-per-read
result.  Prints the number of launches whose SFS differ, and the first differences.
  [SVDSS_LIB=...] [SVDSS_BS=1] python tools/seg_stress.py [threads] [repeats] [segments]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(threads=6, repeats=40, segments=8, n_reads=1200, read_len=3000, seed=91):
    import svdss_amd
    from tests.common import small_workload
    ref, hap, svs, flat, offs = small_workload(seed=seed, n_reads=n_reads, read_len=read_len, ref_lens=(400000, 150000), n_svs=12)
    ix = svdss_amd.FMDIndex.build(ref).to_device(0)
    os.environ["SVDSS_SEGMENTS"] = "1"
    pp = svdss_amd.PingPong(ix, assemble=False)
    want = pp.ping_pong_search(flat, offs)
    pp.close()
    os.environ["SVDSS_SEGMENTS"] = str(segments)
    bad, first = [0] * threads, [None] * threads

    def work(t):
        q = svdss_amd.PingPong(ix, assemble=False)
        for rep in range(repeats):
            got = q.ping_pong_search(flat, offs)
            same = (np.array_equal(got.counts, want.counts) and np.array_equal(got.qs, want.qs) and
                    np.array_equal(got.len, want.len) and np.array_equal(got.n_ext, want.n_ext))
            if not same:
                bad[t] += 1
                if first[t] is None:
                    if np.array_equal(got.counts, want.counts):
                        d = np.nonzero((got.qs != want.qs) | (got.len != want.len))[0]
                        first[t] = (rep, "records", [(int(i), int(got.qs[i]), int(got.len[i]), int(want.qs[i]), int(want.len[i])) for i in d[:4]])
                    else:
                        d = np.nonzero(got.counts != want.counts)[0]
                        first[t] = (rep, "counts", [(int(i), int(got.counts[i]), int(want.counts[i])) for i in d[:4]])
        q.close()

    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return sum(bad), threads * repeats, [f for f in first if f is not None]


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]] + [6, 40, 8][len(sys.argv) - 1:]
    nbad, n, first = run(a[0], a[1], a[2])
    print(f"lib {os.environ.get('SVDSS_LIB', 'default')} BS={os.environ.get('SVDSS_BS', '0')}: {nbad} of {n} launches of {a[2]} segments x {a[0]} threads differ from the "
          f"one-lane-per-read result", first[:3])
