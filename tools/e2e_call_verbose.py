import os, sys, subprocess, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
from tools import e2e_call as EC
work = "/tmp/e2ecv"
fa, cbam, svs, het, recs, hdr, _ = EC.write_dataset(work, 30_000_000, 200, 30, 15000, het_every=2, max_len=2000)
exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
fmd = os.path.join(work, "ref.fmd")
subprocess.run([exe, "index", "-d", fa, "-o", fmd], check=True, capture_output=True)
sfs = os.path.join(work, "s.txt")
for rep in range(2):
    t0 = time.perf_counter()
    with open(sfs, "wb") as f:
        r = subprocess.run([exe, "search", "--index", fmd, "--bam", cbam, "--verbose"], stdout=f, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SVDSS_DEBUG="1"))
    print("search wall", round(time.perf_counter() - t0, 3))
    print("\n".join(l for l in r.stderr.splitlines() if "debug" in l or "[search]" in l or "bam_reader" in l)[-1500:])
    t0 = time.perf_counter()
    r = subprocess.run([exe, "call", "--reference", fa, "--bam", cbam, "--sfs", sfs, "--threads", "16", "--min-sv-length", "50", "--verbose"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    print("call wall", round(time.perf_counter() - t0, 3))
    print("\n".join(l for l in r.stderr.splitlines() if "[time]" in l))
t0 = time.perf_counter()
subprocess.run([exe, "--version"], capture_output=True)
print("SVDSS --version wall", round(time.perf_counter() - t0, 3))
