#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# Evidence set of a round, all from ONE build (run on the GPU box; .git does not travel, so the commit is passed in):
#   gpurun -- 'bash tools/_profile_round.sh r03 <commit> [quick]'
# bench lines + rocprofv3 kernel traces at 1M (the metric's size) and 500k (C3), fabric traffic (PMC) of the hot kernels,
# PMC passes over the search and the recurrence kernels, the per-wave timeline of a recurrence step, and the full-oracle
# parity runs at 1M and 500k, the VertexFrequencyCluster line at 1M and the per-rank compute of the sharded driver.  Every file carries the commit; copy gpurun_out/prof_<tag>/* to profiles/.
tag=${1:-r06}; commit=${2:-unknown}; quick=${3:-}
out=gpurun_out/prof_$tag; mkdir -p $out/pmc; export TMPDIR=/tmp
stamp() { echo "# commit $commit, $(date -u +%Y-%m-%dT%H:%MZ), $(rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//')"; }
cpu=""; [ "$quick" = "cpufull" ] && cpu="--cpu-full"   # (the 27-minute CPU protocol only when asked for)
for n in 1000000 500000; do
  c=$cpu; [ $n != 1000000 ] && c="--cpu-sample 0"
  python bench.py --cells $n $c 2>$out/bench_$n.err > $out/bench_$n.json
  python - $out/bench_$n.json $commit <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d["commit"] = sys.argv[2]
open(sys.argv[1], "w").write(json.dumps(d) + "\n")
PY
  (cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/$out/trace_$n -o t -- python $OLDPWD/bench.py --cells $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-input --no-extra > $OLDPWD/$out/trace_${n}_stdout.log 2>&1)
  db=$(ls $out/trace_$n/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && { stamp; echo "# rocprofv3 --kernel-trace --stats -- python bench.py --cells $n --steps 3 --warmup 1 --cpu-sample 0 --no-host-input"; python tools/rocpd_summary.py $db; } > $out/kernel_stats_$n.md
  rm -rf $out/trace_$n $out/trace_${n}_stdout.log   # (the rocpd database is ~50 MB; gpurun_out is capped at 64 MiB)
  python tools/pmc_traffic.py --cells $n --out $out/pmc/traffic.json > /dev/null 2>&1
done
python - $out/pmc/traffic.json $commit <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for v in d.values():
    if isinstance(v, dict): v["commit"] = sys.argv[2]
json.dump(d, open(sys.argv[1], "w"), indent=1, sort_keys=True)
PY
{ stamp; bash tools/pmc_knn.sh gpurun_out/pmc_knn_$tag 1000000 2>&1 | grep -v "^pass"; } > $out/pmc/knn16_pmc_summary.txt
{ stamp; bash tools/pmc_spmm.sh gpurun_out/pmc_spmm_$tag 1000000 0 2>&1 | grep -v "^pass\|^saved"; } > $out/pmc/spmm_pmc_summary.txt
# the search kernel's L2 counters ride in traffic.json too (hit / miss / fabric reads / L1->L2 read latency per launch)
python - $out/pmc/knn16_pmc_summary.txt $out/pmc/traffic.json $commit <<'PY'
import json, sys
vals = {}
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) >= 4 and f[1] == "nprod1" and f[2] in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_VALU"):
        vals[f[2]] = float(f[3])
d = json.load(open(sys.argv[2]))
if vals:
    rec = dict(vals, commit=sys.argv[3], note="tools/pmc_knn.sh: per launch of the first-pass search kernel at 1M x 50")
    if vals.get("TCC_HIT_sum") is not None and vals.get("TCC_MISS_sum"):
        rec["tcc_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    if vals.get("TCP_TCC_READ_REQ_sum"):
        rec["tcp_tcc_read_latency_cycles_mean"] = vals.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / vals["TCP_TCC_READ_REQ_sum"]
    d["knn16_topk_l2@1000000x50"] = rec
json.dump(d, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
rm -rf gpurun_out/pmc_knn_$tag gpurun_out/pmc_spmm_$tag
{ stamp; python tools/save_graph.py 1000000 /tmp/g1m.pt > /dev/null; for p in 2 1; do echo "## p = $p"; python tools/spmm_stamps.py /tmp/g1m.pt $p 2>/dev/null; done; python tools/spmm_time.py /tmp/g1m.pt 2>/dev/null | grep "tiled p\|lanczos"; } > $out/recurrence_step_timeline.txt
{ stamp; echo "# the lmax estimate: wall time, serial convergence checks (MELD_LANCZOS_SPECULATE=0) against checks overlapped with the next batch: tools/time_lmax_sizes.py"
  for sp in 0 1; do MELD_LANCZOS_SPECULATE=$sp python tools/time_lmax_sizes.py 1000000 500000 200000 2>&1 | grep "^N="; done; } > $out/lmax.txt
{ stamp; for n in 1000000 500000; do echo "## N = $n"; MELD_COMMIT=$commit MELD_CPU_FULL_JSON=$out/cpu_full_size.json python tools/parity_200k.py $n 2>&1 | grep -v amdgpu.ids; done; } > $out/full_oracle_parity.txt
{ stamp; python bench.py --cells 1000000 --steps 2 --warmup 1 --cpu-sample 0 --no-host-input --no-extra --vfc 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('vfc', 'roofline_vfc') if k in d}, indent=1))"; } > $out/vfc_1M.txt
{ stamp; echo "# the wide (lanes = columns) recurrence step, 64 columns at 1M cells: tools/time_wide.py; counters: tools/pmc_kernel.sh cheby_step_wide"
  python tools/time_wide.py 1000000 64 2>&1 | grep -v amdgpu.ids | tail -2
  CMD="python tools/time_wide.py 1000000 64" bash tools/pmc_kernel.sh cheby_step_wide 2>&1 | tail -15
  echo "# where the neighbours lie in the device order (tools/row_locality.py): the share of the nonzeros inside a window of rows"
  python tools/row_locality.py 2>&1 | grep -v amdgpu.ids | tail -8; } > $out/wide_spmm.txt
# the wide kernel's fabric reads / L2 hit rate ride in traffic.json (per launch)
python - $out/wide_spmm.txt $out/pmc/traffic.json $commit <<'PY'
import json, re, sys
vals, n = {}, 1
for line in open(sys.argv[1]):
    m = re.match(r"(\w+)\s+total ([0-9.e+]+) over (\d+) dispatches", line)
    if m:
        vals[m.group(1)] = float(m.group(2)); n = int(m.group(3))
if vals.get("TCC_EA0_RDREQ_sum"):
    d = json.load(open(sys.argv[2]))
    rec = {"commit": sys.argv[3], "launches_averaged": n, "rdreq_per_launch": vals["TCC_EA0_RDREQ_sum"] / n,
           "wrreq_per_launch": vals.get("TCC_EA0_WRREQ_sum", 0.0) / n,
           "bytes_per_launch": (vals["TCC_EA0_RDREQ_sum"] * 128 + vals.get("TCC_EA0_WRREQ_sum", 0.0) * 64) / n,
           "note": "tools/pmc_kernel.sh cheby_step_wide over tools/time_wide.py 1000000 64: RDREQ x 128 B + WRREQ x 64 B per 64-column product"}
    if vals.get("TCC_HIT_sum") is not None and vals.get("TCC_MISS_sum"):
        rec["tcc_hit_rate"] = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
    d["cheby_step_wide@1000000x64cols"] = rec
    json.dump(d, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
{ stamp; echo "# the two passes of the search at 1M x 50 (principal frame): the list-filter pass (K block 0 of every listed pair) and the list-driven search over the thinned lists"
  echo "## product (HIP events around the two launches; tools/knn_only.py)"; python tools/knn_only.py 1000000 4 2>&1 | grep -v amdgpu.ids | tail -3
  echo "## MELD_KNN_TWO_PHASE=0: the round-5 form (one kernel tests and searches, every listed tile staged in full)"; MELD_KNN_TWO_PHASE=0 python tools/knn_only.py 1000000 4 2>&1 | grep -v amdgpu.ids | tail -2
  echo "## MELD_KNN_ROTATE=0: cells as given (no frame, no test)"; MELD_KNN_ROTATE=0 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -1
  echo "## counters of the search kernel behind the filter (MELD_KNN16_STATS): blocks, slow path, appends, where a wave's cycles go"
  MELD_KNN16_STATS=1 python tools/knn_only.py 1000000 1 2>&1 | grep "stats" | head -3
  # (meld_amd/libmeld_hip_prof.so: built BEFORE the call, on the build host -- `bash tools/build_variant.sh prof knn16.hip -DK16_PROFILING`;
  # the object files it links against do not travel to the GPU box)
  if [ -f meld_amd/libmeld_hip_prof.so ]; then
    echo "## filter pass, timing-only ablations of the -DK16_PROFILING build (MELD_KNN_FILTER_ABL: 0 = product, 1 = no staging behind the first step, 2 = no tests, 3 = neither); knn_filter in ms"
    for a in 0 1 2 3; do echo -n "abl $a: "; MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_prof.so MELD_KNN_FILTER_ABL=$a python tools/knn_only.py 1000000 3 2>&1 | grep "knn_filter" | tail -1; done
    echo "## search over the thinned lists, timing-only ablations of the list kernel (MELD_KNN16_ABLATION: 1 = no selection, 8 = MFMAs only with tiles from an L2-hot set, 9 = MFMAs only without tile loads); knn_topk in ms"
    for a in 1 8 9; do echo -n "abl $a: "; MELD_HIP_LIB=$PWD/meld_amd/libmeld_hip_prof.so MELD_KNN16_ABLATION=$a python tools/knn_only.py 1000000 3 2>&1 | grep "knn_filter" | tail -1; done
  else echo "(no profiling build of the library in the tree: ablations skipped)"; fi
  echo "## d = 100 (the reference's default n_pca): 1M x 100, principal frame by the library rotation, the same two passes"
  DIMS=100 python tools/knn_only.py 1000000 3 2>&1 | grep -v amdgpu.ids | tail -3
  DIMS=100 MELD_KNN_ROTATE=0 python tools/knn_only.py 1000000 2 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/list_stats.py 2>&1 | grep -v amdgpu.ids | tail -9
  echo "## what the frame costs (tools/time_rotate.py), what the test could drop (tools/sim_partial.py)"
  python tools/time_rotate.py 2>&1 | grep -v amdgpu.ids | tail -9
  python tools/sim_partial.py 1000000 64 2>&1 | grep -v amdgpu.ids | tail -16
  } > $out/knn_ablation.txt
{ stamp; echo "# per-rank compute of the sharded driver on ONE GPU (stand-in collectives, results wrong by construction): tools/shard_emulate.py"
  for g in 2 4 8; do python tools/shard_emulate.py 1000000 $g 1 2>&1 | grep "^rank\|^collectives\|^lmax" | tail -3; done
  echo "# rank 0 with the recurrences enqueued from C on a REAL one-rank RCCL communicator (RCCL=1: meld_cheby_run_sharded / meld_lanczos_steps_sharded), and the per-step Python loops beside it (RCCL=0)"
  for g in 2 4 8; do for r in 1 0; do echo "## world $g, RCCL=$r"; RCCL=$r python tools/shard_emulate.py 1000000 $g 0 2>&1 | grep "^rank\|^collectives\|^lmax" | tail -3; done; done; } > $out/shard_emulation.txt
{ stamp; echo "# stop rules of the lmax estimate on recorded Lanczos runs (tools/lmax_rule.py): residual rule of the product vs an eigenvalue-error rule"
  python tools/lmax_rule.py 1000000 500000 200000 2>&1 | grep -v amdgpu.ids | grep -v "e+2[0-9][0-9]"; } > $out/lmax_rule.txt
{ stamp; echo "# what a finer pruning granularity than 64 queries x 64 references would compute (tools/sim_granularity.py, the kernel's own rule per piece)"
  python tools/sim_granularity.py 1000000 96 2>&1 | grep -v amdgpu.ids | tail -20; } > $out/knn_granularity.txt
{ stamp; echo "# fuzz of the graph builder against the oracle, graphtools' bandwidth / knn_max options drawn (tools/fuzz_graph.py)"
  FUZZ_OPTIONS=1 FUZZ_N_MAX=8000 timeout 600 python tools/fuzz_graph.py 50 7 2>&1 | grep -v amdgpu.ids | tail -55
  echo "# product only (builds, symmetry, finiteness)"
  FUZZ_OPTIONS=1 FUZZ_NO_ORACLE=1 timeout 300 python tools/fuzz_graph.py 200 11 2>&1 | grep -v amdgpu.ids | grep -v "^ok" | tail -12; } > $out/fuzz.txt
{ stamp; echo "# whole path against the whole oracle (brute-force kNN on all host cores) at other shapes and with graphtools' graph keywords: tools/parity_200k.py"
  DIMS=100 KNN=15 SEED=4 python tools/parity_200k.py 300000 2>&1 | grep -v amdgpu.ids
  DIMS=20 KNN=5 SEED=3 python tools/parity_200k.py 300000 2>&1 | grep -v amdgpu.ids
  DIMS=30 KNN=12 SEED=7 OPTS='{"kernel_symm": "mnn", "theta": 0.4}' python tools/parity_200k.py 270000 2>&1 | grep -v amdgpu.ids; } > $out/parity_shapes.txt
ls -la $out $out/pmc
