#!/bin/bash
mkdir -p gpurun_out/exp13
timeout 600 python -m pytest tests/test_gpu_tiled.py -x -q 2>&1 | tail -3
python tools/save_graph.py 1000000 /tmp/g1m.pt
(timeout 300 python tools/spmm_time.py /tmp/g1m.pt; PT_REPLAN=0 timeout 300 python tools/spmm_time.py /tmp/g1m.pt; python tools/spmm_stamps.py /tmp/g1m.pt 2) 2>&1 | grep -v amdgpu | tee gpurun_out/exp13/time.txt
