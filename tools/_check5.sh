#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
python bench.py --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py
