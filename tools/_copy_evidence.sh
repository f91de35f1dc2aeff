#!/bin/bash
# gpurun_out/prof_<tag>/* (written by tools/_profile_round.sh on the GPU box) -> profiles/<tag>_*: bash tools/_copy_evidence.sh r06
tag=${1:-r06}; src=gpurun_out/prof_$tag
cp $src/bench_1000000.json profiles/${tag}_bench_1M_line.json; cp $src/bench_500000.json profiles/${tag}_bench_500k_line.json
cp $src/kernel_stats_1000000.md profiles/${tag}_bench_1M_kernel_stats.md; cp $src/kernel_stats_500000.md profiles/${tag}_bench_500k_kernel_stats.md
for f in cpu_full_size.json full_oracle_parity.txt fuzz.txt knn_ablation.txt knn_granularity.txt lmax.txt lmax_rule.txt parity_shapes.txt recurrence_step_timeline.txt shard_emulation.txt vfc_1M.txt wide_spmm.txt; do
  [ -f $src/$f ] && cp $src/$f profiles/${tag}_$f
done
cp $src/pmc/knn16_pmc_summary.txt profiles/pmc/${tag}_knn16_pmc_summary.txt; cp $src/pmc/spmm_pmc_summary.txt profiles/pmc/${tag}_spmm_pmc_summary.txt
cp $src/pmc/traffic.json profiles/pmc/${tag}_traffic.json; cp $src/pmc/traffic.json profiles/pmc/traffic.json
ls profiles | grep "^${tag}_" | wc -l
