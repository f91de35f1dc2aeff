timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15
