#!/bin/bash
# round 5: the call side's throughput when a call takes the sub-clusters of G steps at once (3 batches in flight)
mkdir -p gpurun_out
{
for nc in 3395 6790 10185; do
  for T in 2 3; do
    timeout 900 python tools/call_dp_concurrent.py $T 4 $nc 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
} > gpurun_out/r05_call_group.txt 2>&1
cat gpurun_out/r05_call_group.txt
