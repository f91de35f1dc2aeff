export MELD_KNN_TWO_PHASE_EE=1
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
for a in 0 2 3; do echo "== abl $a"; MELD_KNN_FILTER_ABL=$a python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | grep knn_filter | tail -1; done
timeout 600 python -m pytest tests/test_gpu_partial_search.py tests/test_gpu_parity.py -x -q -k "frame or partial" 2>&1 | tail -2
