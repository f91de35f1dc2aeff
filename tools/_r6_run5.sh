echo "== two-phase, B = list kernel"; python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -3
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
echo "== two-phase, B = EE pair kernel"; MELD_KNN_TWO_PHASE_EE=1 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -3
MELD_KNN_TWO_PHASE_EE=1 MELD_KNN16_STATS=1 python tools/knn_only.py 1000000 1 2>&1 | grep -v amdgpu.ids | grep "stats" | head -3
