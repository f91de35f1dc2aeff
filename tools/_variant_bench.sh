#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# bench stage times with variant builds of the library: bash tools/_variant_bench.sh s8 s16 ...
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
echo "== base"; python bench.py --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py
for v in "$@"; do
  cp meld_amd/libmeld_hip_$v.so meld_amd/libmeld_hip.so
  echo "== $v"; python bench.py --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py
done
cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
