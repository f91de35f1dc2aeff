timeout 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -2
python tools/time_host_input.py 1000000 2>&1 | grep -v amdgpu.ids
