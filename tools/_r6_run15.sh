export MELD_DEV=1; export TMPDIR=/tmp
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o t -- python $OLDPWD/bench.py --cells 1000000 --dims 100 --steps 3 --warmup 1 --cpu-sample 0 --no-host-input --no-extra > /tmp/trg.log 2>&1)
f=$(ls /tmp/trg/*/*_kernel_trace.csv /tmp/trg/*_kernel_trace.csv 2>/dev/null | head -1); echo $f
python tools/step_gaps.py $f 100 | cut -c1-220
