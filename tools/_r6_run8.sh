export MELD_DEV=1
for n in 300000 400000; do for t in 1 2; do echo "== N $n two_phase=$t"; MELD_KNN_TWO_PHASE=$t python tools/knn_only.py $n 3 2>&1 | grep -v amdgpu.ids | grep "knn_topk'" | tail -1; done; done
for g in 8 2; do for t in 1 2; do echo "== shard world $g two_phase=$t"; MELD_KNN_TWO_PHASE=$t python tools/shard_emulate.py 1000000 $g 1 2>&1 | grep "^rank" | tail -1; done; done
echo "== wide panels"; python tools/time_wide.py 1000000 64 2>&1 | grep -v amdgpu.ids | tail -2
MELD_WIDE_PANELS=0 python tools/time_wide.py 1000000 64 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_cluster.py -x -q 2>&1 | tail -2
