"""Developer probe: `SVDSS index` on a FASTA of the GRCh38 primary lengths (60-column lines), stage marks (SVDSS_DEBUG)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from svdss_amd import synth
work = "/tmp/ixwg"
os.makedirs(work, exist_ok=True)
ref = synth.make_reference(list(bench.GRCH38_PRIMARY), seed=11)
lut = np.frombuffer(b"NACGTN", dtype=np.uint8)
fa = os.path.join(work, "wg.fa")
t0 = time.time()
with open(fa, "wb") as f:
    for i, c in enumerate(ref):
        f.write(b">c%d\n" % (i + 1))
        a = lut[c]
        n60 = len(a) // 60 * 60
        body = np.concatenate([a[:n60].reshape(-1, 60), np.full((n60 // 60, 1), 10, np.uint8)], axis=1)
        f.write(body.tobytes())
        if n60 < len(a):
            f.write(a[n60:].tobytes() + b"\n")
print("fasta written", round(time.time() - t0, 1), "s", os.path.getsize(fa), flush=True)
del ref
exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
t0 = time.time()
r = subprocess.run([exe, "index", "-d", fa, "-o", os.path.join(work, "wg.fmd")], capture_output=True, text=True, env=dict(os.environ, SVDSS_DEBUG="1"))
print("index wall", round(time.time() - t0, 2), "s rc", r.returncode)
print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[index]")))
