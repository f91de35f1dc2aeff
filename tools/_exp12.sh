#!/bin/bash
mkdir -p gpurun_out/exp12
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --cpu-sample 0 --no-host-input --stages 2>gpurun_out/exp12/bench.err | tee gpurun_out/exp12/bench.json | python tools/_benchline.py
