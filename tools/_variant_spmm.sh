#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# recurrence step time at several sizes with variant builds of the library: bash tools/_variant_spmm.sh nont ...
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
for v in base "$@"; do
  [ $v != base ] && cp meld_amd/libmeld_hip_$v.so meld_amd/libmeld_hip.so
  for n in 250000 500000 1000000; do echo "== $v N=$n"; python tools/spmm_compare.py $n 2>&1 | grep "tiled p"; done
  cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
done
