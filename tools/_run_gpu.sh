timeout 900 python -m pytest tests/test_gpu_cluster.py -x -q -m gpu 2>&1 | tail -2
python tools/time_vfc.py 600 2>&1 | grep -v amdgpu.ids | tail -1
