export MELD_DEV=1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "row_shard" 2>&1 | tail -3
