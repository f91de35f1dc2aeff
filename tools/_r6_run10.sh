export MELD_DEV=1
DIMS=100 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -3
DIMS=80 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "principal_frame" 2>&1 | tail -1
