python -m pytest tests/test_gpu_api.py -x -q -k "mnn" 2>&1 | tail -15
