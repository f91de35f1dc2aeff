#!/bin/bash
# base vs the timing-only 16-bit-index variant of the recurrence kernels
cp meld_amd/libmeld_hip.so /tmp/libmeld_hip_base.so
for v in base idx16; do
  [ $v != base ] && cp meld_amd/libmeld_hip_$v.so meld_amd/libmeld_hip.so
  for n in 1000000 500000; do echo "== $v"; python tools/idx16_probe.py $n 2>&1 | grep "N="; done
  cp /tmp/libmeld_hip_base.so meld_amd/libmeld_hip.so
done
