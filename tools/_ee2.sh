#!/bin/bash
python tools/time_rotate.py 2>&1 | tail -9
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
echo "== base"; python tools/knn_only.py 1000000 4 2>&1 | tail -1
cp meld_amd/libmeld_hip_w4.so meld_amd/libmeld_hip.so
echo "== 4 waves per SIMD"; python tools/knn_only.py 1000000 4 2>&1 | tail -1
cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
