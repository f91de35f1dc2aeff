#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# search kernel time with variant builds: bash tools/_vk.sh v1 v2 ...
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
echo "== base"; python tools/knn_only.py 1000000 4 2>&1 | tail -1
for v in "$@"; do
  cp meld_amd/libmeld_hip_$v.so meld_amd/libmeld_hip.so
  echo "== $v"; python tools/knn_only.py 1000000 4 2>&1 | tail -1
done
cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
