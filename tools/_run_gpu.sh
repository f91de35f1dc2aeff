mkdir -p gpurun_out/r2x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r2x/prof_stdout.log 2>&1
python tools/rocpd_summary.py /tmp/prof/bench_results.db > gpurun_out/r2x/kernel_stats.md 2>&1
head -12 gpurun_out/r2x/kernel_stats.md | cut -c1-200
python bench.py --steps 5 --warmup 1 --stages > gpurun_out/r2x/bench_full.json 2> gpurun_out/r2x/bench_full.err
cut -c1-300 gpurun_out/r2x/bench_full.json; grep -o '"stages.*' gpurun_out/r2x/bench_full.json | cut -c1-700
