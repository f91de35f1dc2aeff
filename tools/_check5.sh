#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
python bench.py --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py
