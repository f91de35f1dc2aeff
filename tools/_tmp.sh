cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 > gpurun_out/line.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > /tmp/prof_stdout.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/kernel_stats.md
bash tools/pmc_knn.sh /tmp/pmc_knn3 1000000 > gpurun_out/pmc_summary_v3.txt 2>&1
