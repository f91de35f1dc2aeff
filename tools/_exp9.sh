#!/bin/bash
mkdir -p gpurun_out/exp9
python tools/save_graph.py 1000000 /tmp/g1m.pt
for m in 0 1 3 4 5 7; do PT_MASK=$m timeout 300 python tools/spmm_time.py /tmp/g1m.pt; done 2>&1 | grep "tiled p" | tee gpurun_out/exp9/time.txt
