#!/bin/bash
mkdir -p gpurun_out/exp10
python tools/save_graph.py 1000000 /tmp/g1m.pt
(python tools/spmm_stamps.py /tmp/g1m.pt 2; PT_MASK=7 python tools/spmm_stamps.py /tmp/g1m.pt 2) 2>&1 | grep -v amdgpu | tee gpurun_out/exp10/stamps.txt
