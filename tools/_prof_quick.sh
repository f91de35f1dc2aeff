#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# quick kernel trace of the 1M bench (run on the GPU box): bash tools/_prof_quick.sh [rows-to-show]
out=gpurun_out/prof_quick; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/trace -o t -- python $OLDPWD/bench.py --cells ${CELLS:-1000000} --steps 3 --warmup 1 --cpu-sample 0 --no-host-input --no-extra > $OLDPWD/$out/stdout.log 2>&1)
db=$(ls $out/trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db > $out/kernel_stats.md
rm -rf $out/trace
head -${1:-14} $out/kernel_stats.md | cut -c1-175
