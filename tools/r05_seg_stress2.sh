#!/bin/bash
mkdir -p gpurun_out /tmp/hz
cp svdss_amd/libsvdss_hip_hazard.so /tmp/hz/libsvdss_hip.so
{
echo "== round 4's store (no wait states behind it)"
LD_LIBRARY_PATH=/tmp/hz:$LD_LIBRARY_PATH timeout 900 python tools/seg_stress_bam.py 8 2>&1 | grep -v amdgpu.ids | tail -4
echo "== this tree"
timeout 900 python tools/seg_stress_bam.py 8 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r05_seg_stress2.txt 2>&1
cat gpurun_out/r05_seg_stress2.txt
