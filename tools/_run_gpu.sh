{
MELD_KNN16_ABLATION=9 python tools/knn_only.py 1000000 2
MELD_KNN16_ABLATION=8 python tools/knn_only.py 1000000 2
MELD_KNN16_ABLATION=3 python tools/knn_only.py 1000000 2
MELD_KNN16_ABLATION=1 python tools/knn_only.py 1000000 2
python tools/knn_only.py 1000000 3
} 2>&1 | grep -v amdgpu.ids | cut -c1-200
