#!/bin/bash
# search-stage time with variant builds of the library (tools/build_variant.sh): bash tools/_vk.sh [N] v1 v2 ...   ("base" = the in-tree build)
n=$1; shift
for v in "$@"; do
  lib=meld_amd/libmeld_hip_$v.so; [ "$v" = base ] && lib=meld_amd/libmeld_hip.so
  echo "== $v"; MELD_HIP_LIB=$PWD/$lib python tools/knn_only.py $n 3 2>&1 | grep -v amdgpu.ids | tail -3
done
