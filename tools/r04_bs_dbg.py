import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svdss_amd
from tests import oracle_lib as O
from tests.common import small_workload
os.environ["SVDSS_KMER"] = os.environ.get("DBG_K", "8")
ref, hap, svs, flat, offs = small_workload(seed=91, n_reads=400, read_len=1500, ref_lens=(150000,))
reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(400)]
reads[7][40] = 5
reads[11] = reads[11][:99]; reads[12] = reads[12][:100]; reads[13] = reads[13][:101]
reads[14] = np.concatenate([reads[14], reads[15], reads[16], reads[17]])
reads = [r for r in reads if len(r) >= 100]
ix = svdss_amd.FMDIndex.build(ref).to_device(0)
fm = O.OracleFMD.build(ref)
n_bad = 0
for assemble in (True, False):
    for bs in ("1", "0"):
        for step in (len(reads), 57, 13):
            for a in range(0, len(reads), step):
                sub = reads[a:a + step]
                f, o = svdss_amd.pack_reads(sub)
                c, q, l, e = fm.search_batch(f, o, assemble)
                os.environ.update(SVDSS_BS=bs)
                os.environ.pop("SVDSS_SEGMENTS", None)
                pp = svdss_amd.PingPong(ix, assemble=assemble)
                got = pp.ping_pong_search(f, o)
                pp.close()
                ok = (got.counts == c).all() and (got.n_ext == e).all() and (got.qs == q).all() and (got.len == l).all()
                if not ok:
                    n_bad += 1
                    msg = "counts differ"
                    if (got.counts == c).all():
                        bad = np.flatnonzero((got.qs != q) | (got.len != l))
                        first = np.concatenate([[0], np.cumsum(c)])
                        r = int(np.searchsorted(first, bad[0], side="right") - 1)
                        msg = f"first bad record {bad[0]} (read {a + r}, len {len(sub[r])}): got ({got.qs[bad[0]]},{got.len[bad[0]]}) want ({q[bad[0]]},{l[bad[0]]}); {len(bad)} bad; n_ext equal {bool((got.n_ext == e).all())}"
                    if n_bad < 12:
                        print("assemble", assemble, "BS", bs, "batch", a, step, msg, flush=True)
print("bad batches:", n_bad)
