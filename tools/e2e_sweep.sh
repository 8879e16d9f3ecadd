# developer sweep: one synthetic BAM (E2E_REPEAT x 172,000 reads), `SVDSS search` under several pipeline settings
export PYTHONPATH=.
export SVDSS_DEBUG=1
E2E_REPEAT=${E2E_REPEAT:-6} timeout 1200 python tools/e2e_search.py 64444167 172000 15000 /tmp/e2e --verbose 2>&1 | grep -v amdgpu.ids | grep "^{" 
run() {
  echo "== $*"
  env "$@" ./svdss_amd/SVDSS search --index /tmp/e2e/ref.fmd --bam /tmp/e2e/reads.bam --noputative --verbose 2>&1 >/tmp/e2e/out_$N.sfs | grep "stage busy\|records read\|device at\|rror\|parser waited"
  md5sum /tmp/e2e/out_$N.sfs | cut -c1-12
}
N=1 run X=1
N=2 run SVDSS_BAM_SLAB_KB=16384
N=3 run SVDSS_BAM_SLAB_KB=8192 SVDSS_BAM_AHEAD=24
N=4 run SVDSS_BAM_SLAB_KB=8192 SVDSS_BAM_AHEAD=32 SVDSS_SEARCH_FEEDERS=4
