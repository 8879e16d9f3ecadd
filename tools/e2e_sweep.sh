# developer sweep: one synthetic BAM (E2E_REPEAT x 172,000 reads), `SVDSS search` (text to /dev/null) under several
# pipeline settings; prints the streaming time (records read at +t s, after the index restore) per setting
export PYTHONPATH=.
export SVDSS_DEBUG=1
E2E_REPEAT=${E2E_REPEAT:-6} timeout 1200 python tools/e2e_search.py 64444167 172000 15000 /tmp/e2e --verbose 2>&1 | grep -v amdgpu.ids | grep "^{" | cut -c1-300
run() {
  echo "== $*"
  for k in 1 2; do
    env "$@" ./svdss_amd/SVDSS search --index /tmp/e2e/ref.fmd --bam /tmp/e2e/reads.bam --noputative --verbose 2>&1 >/dev/null | grep "records read\|device at\|rror\|parser waited\|chunks inflated" | cut -c30-200
  done
}
run X=1
run SVDSS_GPU_INFLATE=100
run SVDSS_GPU_INFLATE=90
run SVDSS_GPU_INFLATE=80
run SVDSS_GPU_INFLATE=70
run SVDSS_BAM_AHEAD=24
run SVDSS_BAM_AHEAD=32 SVDSS_BAM_SLAB_KB=16384
