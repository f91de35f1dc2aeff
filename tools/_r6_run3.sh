echo "== default (two-phase)"; python tools/knn_only.py 1000000 4 2>&1 | grep -v amdgpu.ids | tail -2
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
echo "== MELD_KNN_TWO_PHASE=0"; MELD_KNN_TWO_PHASE=0 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -1
echo "== 500k"; python tools/knn_only.py 500000 3 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_partial_search.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "frame or partial or 50k_config or odd_number or tile_pruning or direct_step or knn_max or symmetrisation or oracle_digest or sharded_recurrences" 2>&1 | tail -8
