#!/bin/bash
# round 4, first GPU batch: (E1) random-line rate against buffer size, (E3) issue rate of half-empty wavefronts,
# (E2) POA kernel built for 5 / 6 / 8 wavefronts per SIMD, alone and inside the bench pipeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04a; mkdir -p $O
( for g in 0.125 0.25 0.5 1 2 4 8 16 32 64; do ./tools/random_read_probe $g 128; done ) > $O/random_read_cliff.txt 2>&1
./tools/issue_probe > $O/issue_probe.txt 2>&1
for v in "" _poa5 _poa6 _poa8; do
  export SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip$v.so
  echo "== libsvdss_hip$v.so"
  timeout 300 python tools/call_dp_concurrent.py 1 3
  timeout 300 python tools/call_dp_concurrent.py 3 4
  timeout 300 python tools/poa_long_probe.py 16 2600 30
done > $O/poa_occ.txt 2>&1
for v in "" _poa5 _poa8; do
  SVDSS_LIB=$PWD/svdss_amd/libsvdss_hip$v.so timeout 600 python bench.py --steps 10 --warmup 3 --no-e2e > $O/bench$v.json 2> $O/bench$v.err
done
tail -n 3 $O/poa_occ.txt; head -c 600 $O/bench.json
