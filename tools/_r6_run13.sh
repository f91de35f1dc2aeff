export MELD_DEV=1; export TMPDIR=/tmp; mkdir -p gpurun_out/r6d
(cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/tr100 -o t -- python $OLDPWD/bench.py --cells 1000000 --dims 100 --steps 2 --warmup 1 --cpu-sample 0 --no-host-input --no-extra > /tmp/tr100.log 2>&1)
db=$(ls /tmp/tr100/*.db 2>/dev/null | head -1); python tools/rocpd_summary.py $db | head -24 | cut -c1-200
