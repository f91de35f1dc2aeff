#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_cluster.py tests/test_gpu_tiled.py -x -q 2>&1 | tail -8
