"""Developer probe: `SVDSS search` on the bench's end-to-end BAM with different shares of the BGZF chunks inflated by the
host's libdeflate workers beside the GPU (SVDSS_GPU_INFLATE)."""
import os, re, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import e2e_search as E
work = "/tmp/e2e_ab"
os.makedirs(work, exist_ok=True)
rng = np.random.default_rng(1)
ref = rng.integers(0, 4, size=64444167, dtype=np.uint8)
fa = os.path.join(work, "ref.fa")
with open(fa, "wb") as f:
    f.write(b">chrS\n"); f.write(np.frombuffer(b"ACGT", dtype=np.uint8)[ref].tobytes()); f.write(b"\n")
bam = os.path.join(work, "reads.bam")
E.write_bam(bam, "chrS", ref, 172000, 15000, repeat=6)
exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
subprocess.run([exe, "index", "-d", fa, "-o", os.path.join(work, "ref.fmd")], check=True, capture_output=True)
for mode in ("101", "100", "90", "80", "101", "100"):
    env = dict(os.environ, SVDSS_DEBUG="1", SVDSS_GPU_INFLATE=mode)
    t0 = time.perf_counter()
    r = subprocess.run([exe, "search", "--index", os.path.join(work, "ref.fmd"), "--bam", bam, "--noputative", "--verbose"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, check=True, env=env)
    wall = time.perf_counter() - t0
    t_ix = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1))
    t_end = float(re.search(r"SFS written at \+([0-9.]+) s", r.stderr).group(1))
    g = re.search(r"(\d+) chunks inflated on the GPU", r.stderr)
    print(f"SVDSS_GPU_INFLATE={mode}: streaming {t_end - t_ix:.2f} s, whole process {wall:.2f} s, chunks on the GPU {g.group(1) if g else 0}", flush=True)
