#!/bin/bash
# round-3 experiment 1: baseline bench + structure stats + cache-policy variants of the recurrence stream
mkdir -p gpurun_out/exp1
python bench.py --cpu-sample 0 --no-host-input --stages 2>gpurun_out/exp1/bench.err | tee gpurun_out/exp1/bench.json | python tools/_benchline.py
python tools/save_graph.py 1000000 /tmp/g1m.pt
python tools/block_stats.py /tmp/g1m.pt 2>&1 | tee gpurun_out/exp1/block_stats.txt
for v in "" nt_none nt_sc1 nt_sc0sc1 nt_ntsc1 nt_ntsc0sc1; do
  if [ -z "$v" ]; then python tools/spmm_time.py /tmp/g1m.pt; else MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_$v.so python tools/spmm_time.py /tmp/g1m.pt; fi
done 2>&1 | grep -v "^$" | tee gpurun_out/exp1/variants.txt
