timeout 900 python -m pytest tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -15
