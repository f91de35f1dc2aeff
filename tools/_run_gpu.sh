mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3a/prof_stdout.log 2>&1
python tools/rocpd_summary.py /tmp/prof/bench_results.db > gpurun_out/r3a/kernel_stats.md 2>&1
head -8 gpurun_out/r3a/kernel_stats.md | cut -c1-200
python bench.py --steps 5 --warmup 1 --stages > gpurun_out/r3a/bench_full.json 2> gpurun_out/r3a/bench_full.err
cut -c1-250 gpurun_out/r3a/bench_full.json
bash tools/pmc_knn.sh /tmp/pmc_knn 1000000 > gpurun_out/r3a/pmc_summary.txt 2>&1
tail -60 gpurun_out/r3a/pmc_summary.txt | grep "nprod1" | cut -c1-150
