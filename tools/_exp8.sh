#!/bin/bash
mkdir -p gpurun_out/exp8
timeout 600 python -m pytest tests/test_gpu_tiled.py -x -q -k "awkward or oracle" 2>&1 | tail -3
python tools/save_graph.py 1000000 /tmp/g1m.pt
(timeout 300 python tools/spmm_time.py /tmp/g1m.pt; PT_MASK=4 timeout 300 python tools/spmm_time.py /tmp/g1m.pt; MELD_SPMM_FOLD=0 timeout 300 python tools/spmm_time.py /tmp/g1m.pt) 2>&1 | grep "tiled p" | tee gpurun_out/exp8/time.txt
