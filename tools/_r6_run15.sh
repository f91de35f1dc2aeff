export MELD_DEV=1; export TMPDIR=/tmp
for n in 1000000 500000; do
(cd /tmp; rm -rf /tmp/trg; rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o t -- python $OLDPWD/bench.py --cells $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-input --no-extra > /tmp/trg.log 2>&1)
f=$(ls /tmp/trg/*/*_kernel_trace.csv /tmp/trg/*_kernel_trace.csv 2>/dev/null | head -1); echo "== N=$n"
python tools/step_gaps.py $f 20 | cut -c1-200 | head -16
done
