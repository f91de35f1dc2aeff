#!/bin/bash
mkdir -p gpurun_out/exp5
python tools/save_graph.py 1000000 /tmp/g1m.pt
for v in abl1 abl2 abl4 abl8 abl16 abl31; do
  MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_$v.so timeout 300 python tools/spmm_time.py /tmp/g1m.pt
done 2>&1 | grep "tiled p" | tee gpurun_out/exp5/time.txt
