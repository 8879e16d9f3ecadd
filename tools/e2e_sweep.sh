# developer sweep: one synthetic BAM (E2E_REPEAT x 172,000 reads), `SVDSS search` (text to /dev/null) under several
# pipeline settings; prints "index ready at / everything written at" (seconds since process start) per run
export PYTHONPATH=.
E2E_REPEAT=${E2E_REPEAT:-6} timeout 1200 python tools/e2e_search.py 64444167 172000 15000 /tmp/e2e 2>&1 | grep -v amdgpu.ids | grep "^{" | cut -c1-100
run() {
  echo "== $*"
  for k in 1 2 3; do
    env "$@" ./svdss_amd/SVDSS search --index /tmp/e2e/ref.fmd --bam /tmp/e2e/reads.bam --noputative --verbose 2>&1 >/dev/null | grep "records read\|device at" | sed 's/.* at +//' | tr '\n' ' '; echo
  done
}
run X=1
run SVDSS_BAM_AHEAD=12
run SVDSS_BAM_AHEAD=20
run SVDSS_BAM_AHEAD=24 SVDSS_BAM_SLAB_KB=16384
run SVDSS_SEARCH_FEEDERS=4
run SVDSS_SEARCH_FEEDERS=8
run SVDSS_GPU_INFLATE=90
