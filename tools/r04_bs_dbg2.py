import os, subprocess, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import svdss_amd
from tests.common import small_workload, BIN
from tests import test_bam_device_gpu as T
ref, hap, svs, flat, offs = small_workload(seed=91, n_reads=400, read_len=1500, ref_lens=(150000,))
ix = svdss_amd.FMDIndex.build(ref).to_device(0)
reads = [flat[offs[i]:offs[i + 1]].copy() for i in range(400)]
reads[7][40] = 5
reads[11] = reads[11][:99]; reads[12] = reads[12][:100]; reads[13] = reads[13][:101]
reads[14] = np.concatenate([reads[14], reads[15], reads[16], reads[17]])
names = [f"m64/{(i * 37) % 400}/ccs" for i in range(400)]
names[21] = "x"
rng = np.random.default_rng(23)
recs = []
for rep in range(8):
    r, _ = T._records([f"{n}/{rep}" for n in names], reads, rng, decoys=(rep % 2 == 0))
    recs += r
os.makedirs("/tmp/dbg2", exist_ok=True)
bam = "/tmp/dbg2/reads.bam"
open(bam, "wb").write(T._bgzf_levels(T._raw_bam([("chr1", 150000)], recs), rng, block=60000))
fmd = "/tmp/dbg2/ref.fmd"
ix.save(fmd)
outs = {}
base = {"SVDSS_BAM_BATCH_MB": "1", "SVDSS_BAM_SLAB_KB": "64"}
for name, env in [("dev BS1 table off", dict(base, SVDSS_BS="1", SVDSS_BS_DBG="1")),
                  ("dev BS1 seg 2", dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="2")),
                  ("dev BS1 seg 4", dict(base, SVDSS_BS="1", SVDSS_SEGMENTS="4")),
                  ("dev BS1 2 feeders", dict(base, SVDSS_BS="1", SVDSS_SEARCH_FEEDERS="2"))]:
    for rep in range(4):
        r = subprocess.run([BIN, "search", "--index", fmd, "--bam", bam, "--threads", "4", "--bsize", "100", "--verbose"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        k = [ln for ln in r.stderr.splitlines() if "k-mer table" in ln]
        print(name, rep, r.returncode, hashlib.md5(r.stdout.encode()).hexdigest()[:10], len(r.stdout), k[-1][-40:] if k else "", flush=True)
        outs[(name, rep)] = r.stdout
a, b = outs[("dev BS0", 0)].splitlines(), outs[("dev BS1", 0)].splitlines()
d = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
print(len(a), len(b), d[:6])
