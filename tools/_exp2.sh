#!/bin/bash
mkdir -p gpurun_out/exp2
python tools/save_graph.py 1000000 /tmp/g1m.pt
for v in "" abl1 abl2 abl3; do for m in 0 4; do
  if [ -z "$v" ]; then PT_MASK=$m python tools/spmm_time.py /tmp/g1m.pt; else PT_MASK=$m MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_$v.so python tools/spmm_time.py /tmp/g1m.pt; fi
done; done 2>&1 | grep "tiled p" | tee gpurun_out/exp2/ablate.txt
