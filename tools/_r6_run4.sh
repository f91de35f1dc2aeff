MELD_KNN16_STATS=1 python tools/knn_only.py 1000000 1 2>&1 | grep -v amdgpu.ids | grep "stats\|pairs\|knn_topk" | head -12
