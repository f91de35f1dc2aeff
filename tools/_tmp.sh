python -m pytest tests/test_gpu_parity.py -x -q -k "awkward" 2>&1 | tail -15
