"""round 4: `SVDSS search --bam` end to end on the bench's BAM (1,032,000 x 15 kb reads, chr20-length index): the device
path (records on the GPU) against the host path, and a few settings of the device path.
  python tools/r04_e2e.py [reads] [workdir]"""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import e2e_search as E  # noqa: E402


def run_search(exe, fmd, bam, env):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "search", "--index", fmd, "--bam", bam, "--noputative", "--verbose"], stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, text=True, env=dict(os.environ, SVDSS_DEBUG="1", **env))
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    t_ix = float(re.search(r"on the device at \+([0-9.]+) s", r.stderr).group(1))
    m = re.search(r"(\d+) records read, (\d+) SFS written at \+([0-9.]+) s", r.stderr)
    n, n_sfs, t_end = int(m.group(1)), int(m.group(2)), float(m.group(3))
    dev = re.search(r"device path: .*", r.stderr)
    return {"reads": n, "sfs": n_sfs, "index_resident_s": round(t_ix, 3), "streaming_s": round(t_end - t_ix, 3),
            "reads_per_s_streaming": round(n / max(t_end - t_ix, 1e-9)), "whole_process_s": round(wall, 3),
            "stages": dev.group(0) if dev else re.search(r"stage busy seconds: .*", r.stderr).group(0),
            "device_stages": (re.search(r"device batches, seconds summed: .*", r.stderr) or [None])[0]}


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1032000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/svdss_r04_e2e"
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(1)
    ref = rng.integers(0, 4, size=64444167, dtype=np.uint8)
    fa = os.path.join(work, "chr.fa")
    with open(fa, "wb") as f:
        f.write(b">chrS\n")
        f.write(np.frombuffer(b"ACGT", dtype=np.uint8)[ref].tobytes())
        f.write(b"\n")
    unit = 172000
    bam = os.path.join(work, "reads.bam")
    raw = E.write_bam(bam, "chrS", ref, unit, 15000, repeat=max(1, round(n_reads / unit)))
    exe = os.path.join(ROOT, "svdss_amd", "SVDSS")
    fmd = os.path.join(work, "chr.fmd")
    subprocess.run([exe, "index", "-d", fa, "-o", fmd], check=True, capture_output=True)
    out = {"bam_bytes": os.path.getsize(bam), "inflated_bytes": raw}
    if os.environ.get("R04_ONLY_BUILD"):
        print(json.dumps(out))
        return
    settings = [("host path (SVDSS_BAM_DEVICE=0)", {"SVDSS_BAM_DEVICE": "0"}),
                ("device path, defaults", {}),
                ("device path, defaults (again)", {}),
                ("device path, 6 feeders", {"SVDSS_SEARCH_FEEDERS": "6"}),
                ("device path, 8 feeders", {"SVDSS_SEARCH_FEEDERS": "8"}),
                ("device path, 128 MB batches, 8 feeders", {"SVDSS_SEARCH_FEEDERS": "8", "SVDSS_BAM_BATCH_MB": "128"}),
                ("device path, 512 MB batches", {"SVDSS_BAM_BATCH_MB": "512"}),
                ("device path, 2 feeders", {"SVDSS_SEARCH_FEEDERS": "2"})]
    if os.environ.get("R04_SHORT"):
        settings = [("device path, defaults", {}), ("device path, defaults (2)", {}), ("12 loaders", {"SVDSS_BAM_LOADERS": "12"}), ("12 loaders (2)", {"SVDSS_BAM_LOADERS": "12"}),
                    ("16 loaders, 128 MB", {"SVDSS_BAM_LOADERS": "16", "SVDSS_BAM_BATCH_MB": "128"}), ("4 loaders", {"SVDSS_BAM_LOADERS": "4"}),
                    ("12 loaders, 3 format threads", {"SVDSS_BAM_LOADERS": "12", "SVDSS_FORMAT_THREADS": "3"}),
                    ("device path, 128 MB batches, 8 feeders", {"SVDSS_SEARCH_FEEDERS": "8", "SVDSS_BAM_BATCH_MB": "128"}),
                    ("device path, 128 MB batches, 8 feeders (2)", {"SVDSS_SEARCH_FEEDERS": "8", "SVDSS_BAM_BATCH_MB": "128"}),
                    ("device path, 128 MB batches, 6 feeders", {"SVDSS_SEARCH_FEEDERS": "6", "SVDSS_BAM_BATCH_MB": "128"}),
                    ("device path, 128 MB batches, 4 feeders", {"SVDSS_BAM_BATCH_MB": "128"}),
                    ("device path, 64 MB batches, 8 feeders", {"SVDSS_SEARCH_FEEDERS": "8", "SVDSS_BAM_BATCH_MB": "64"}),
                    ("device path, 3 feeders", {"SVDSS_SEARCH_FEEDERS": "3"})]
    if os.environ.get("R04_ONE"):
        settings = [("defaults", {}), ("defaults (2)", {}), ("defaults (3)", {}), ("6 feeders, 128 MB", {"SVDSS_SEARCH_FEEDERS": "6", "SVDSS_BAM_BATCH_MB": "128"}),
                    ("6 feeders, 128 MB (2)", {"SVDSS_SEARCH_FEEDERS": "6", "SVDSS_BAM_BATCH_MB": "128"}), ("3 feeders", {"SVDSS_SEARCH_FEEDERS": "3"}), ("1 feeder", {"SVDSS_SEARCH_FEEDERS": "1"})]
    if os.environ.get("R04_SMOOTHED"):
        # the BAM `SVDSS smooth` writes (literal-only dynamic Huffman from csrc/deflate.hip), as search sees it in run_svdss
        sm = os.path.join(work, "smoothed.bam")
        t0 = time.perf_counter()
        with open(sm, "wb") as f:
            subprocess.run([exe, "smooth", "--reference", fa, "--bam", bam, "--threads", "16"], check=True, stdout=f, stderr=subprocess.DEVNULL)
        out["smooth_s"] = round(time.perf_counter() - t0, 2)
        print("smooth", out["smooth_s"], "s", flush=True)
        os.remove(bam)
        bam = sm
        settings = [("device path, defaults", {}), ("1 feeder", {"SVDSS_SEARCH_FEEDERS": "1"}), ("2 feeders", {"SVDSS_SEARCH_FEEDERS": "2"}),
                    ("8 feeders, 128 MB", {"SVDSS_SEARCH_FEEDERS": "8", "SVDSS_BAM_BATCH_MB": "128"})]
    for name, env in settings:
        out[name] = run_search(exe, fmd, bam, env)
        print(name, json.dumps(out[name]), flush=True)
    if os.environ.get("R04_ROCPROF"):
        subprocess.run("cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d %s/prof -o search -- %s search --index %s --bam %s --noputative > /dev/null 2> %s/prof.err"
                       % (work, exe, fmd, bam, work), shell=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
