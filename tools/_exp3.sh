#!/bin/bash
mkdir -p gpurun_out/exp3
python tools/save_graph.py 1000000 /tmp/g1m.pt
(PT_NNZ_FULL=39385792 PT_DROP_LOWER_INBLOCK=3906 python tools/spmm_time.py /tmp/g1m.pt
PT_NNZ_FULL=39385792 PT_DROP_LOWER_INBLOCK=3906 MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_abl1.so python tools/spmm_time.py /tmp/g1m.pt
PT_NNZ_FULL=39385792 PT_DROP_LOWER_INBLOCK=3906 PT_MASK=4 python tools/spmm_time.py /tmp/g1m.pt) 2>&1 | grep "tiled p\|dropped" | tee gpurun_out/exp3/drop.txt
