#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# What the round-end checks run on the GPU box: gpurun -- 'bash tools/_run_gpu.sh'
mkdir -p gpurun_out/check
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
python bench.py 2>gpurun_out/check/bench.err | tee gpurun_out/check/bench.json | cut -c1-300
