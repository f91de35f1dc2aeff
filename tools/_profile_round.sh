#!/bin/bash
# Round profile set (run on the GPU box): bench lines + rocprofv3 kernel traces at 1M (the metric's size) and 500k (C3)
# usage: bash tools/_profile_round.sh r02
tag=${1:-r02}; out=gpurun_out/prof_$tag; mkdir -p $out; export TMPDIR=/tmp
for n in 1000000 500000; do
  python bench.py --cells $n 2>$out/bench_$n.err > $out/bench_$n.json
  (cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/trace_$n -o t -- python $OLDPWD/bench.py --cells $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-input > $OLDPWD/$out/trace_${n}_stdout.log 2>&1)
  db=$(ls $out/trace_$n/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/kernel_stats_$n.md
  rm -rf $out/trace_$n   # (the rocpd database is ~50 MB; gpurun_out is capped at 64 MiB)
done
ls -la $out
