#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# tests of the recurrence kernel + one bench line with stage times (run on the GPU box)
mkdir -p gpurun_out/check
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
python bench.py --cpu-sample 0 --no-host-input --stages 2>gpurun_out/check/bench.err | tee gpurun_out/check/bench.json | python tools/_benchline.py
