timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_ranks" 2>&1 | tail -25
