#!/bin/bash
# round 5 (second session): the deflate kernel before / after runs became matches, per kind of quality string
# (tools/deflate_probe.py; libraries: the tree's, the literal-only kernel of the commit before, the tree's with runs switched off)
cd "$(dirname "$0")/.."
for lib in svdss_amd/libsvdss_hipdefl_old.so svdss_amd/libsvdss_hipdefl_noruns.so svdss_amd/libsvdss_hip.so; do
  for kind in random binned hifi ff; do
    echo "== $(basename $lib) $kind"
    SVDSS_LIB=$PWD/$lib python tools/deflate_probe.py 4096 $kind 2>&1 | grep -v "^zlib\|amdgpu.ids" | tail -1
  done
done
