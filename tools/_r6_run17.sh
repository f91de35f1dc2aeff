export MELD_DEV=1
for f in 0 1 1; do echo "== direct $f"; MELD_KNN_FILTER_DIRECT=$f python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | grep "knn_filter" | tail -1; done
