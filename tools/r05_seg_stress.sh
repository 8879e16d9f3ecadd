#!/bin/bash
mkdir -p gpurun_out
{
for lib in svdss_amd/libsvdss_hip_hazard.so svdss_amd/libsvdss_hip.so; do
  for bs in 1 0; do
    SVDSS_LIB=$PWD/$lib SVDSS_BS=$bs timeout 600 python tools/seg_stress.py 6 40 8 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
timeout 600 python -m pytest tests/test_seg_determinism_gpu.py tests/test_sfs_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r05_seg_stress.txt 2>&1
cat gpurun_out/r05_seg_stress.txt
