export MELD_DEV=1
for s in 1 2 3; do echo "== main_slices $s"; MELD_KNN_MAIN_SLICES=$s MELD_KNN_TWO_PHASE=2 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | grep "knn_filter" | tail -1; done
