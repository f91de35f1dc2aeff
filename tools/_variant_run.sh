#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# time tools/knn_ablate.py (product lines) with variant builds of the library: bash tools/_variant_run.sh a3 a4 ...
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
echo "== base"; python tools/knn_ablate.py 2>&1 | grep "product"
for v in "$@"; do
  cp meld_amd/libmeld_hip_$v.so meld_amd/libmeld_hip.so
  echo "== $v"; python tools/knn_ablate.py 2>&1 | grep "product"
done
cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
