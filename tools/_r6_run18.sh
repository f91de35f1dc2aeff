export MELD_DEV=1
MELD_KNN_SAVE_THR=/tmp/thr.pt python tools/knn_only.py 1000000 1 2>&1 | grep -v amdgpu.ids | tail -3
echo "== seeds = final thresholds"
MELD_KNN_SEEDS_FROM=/tmp/thr.pt MELD_KNN16_STATS=1 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | grep "stats\|knn_filter\|pairs" | tail -6
