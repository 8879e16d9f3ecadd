cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
python - <<'PY'
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd())
from tools import e2e_call_wg as W
fa, bam, svs, n, lens = W.write_dataset("/tmp/callwg", 200000, 600)
exe = "svdss_amd/SVDSS"
subprocess.run([exe, "index", "-d", fa, "-o", "/tmp/callwg/ref.fmd"], check=True, capture_output=True)
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([exe, "search", "--index", "/tmp/callwg/ref.fmd", "--bam", bam, "--verbose"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, SVDSS_INDEX_VERBOSE="1", SVDSS_DEBUG="1"))
    print("run", rep, "wall", round(time.time() - t0, 2))
    print(r.stderr[-6000:])
PY
