#!/bin/bash
mkdir -p gpurun_out/exp7
python tools/save_graph.py 1000000 /tmp/g1m.pt
(MELD_SPMM_FOLD=0 timeout 300 python tools/spmm_time.py /tmp/g1m.pt; MELD_SPMM_FOLD=0 PT_MASK=4 timeout 300 python tools/spmm_time.py /tmp/g1m.pt) 2>&1 | grep "tiled p" | tee gpurun_out/exp7/time.txt
