#!/bin/bash
mkdir -p gpurun_out/exp6
python tools/save_graph.py 1000000 /tmp/g1m.pt
timeout 300 python tools/spmm_time.py /tmp/g1m.pt 2>&1 | grep "tiled p" | tee gpurun_out/exp6/time.txt
