#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# PMC passes over the candidate-search kernel (run on the GPU box): tools/pmc_knn.sh <outdir> [N]
# One rocprofv3 run per counter set (counters only, no tracing domains besides --kernel-trace).
out=${1:-gpurun_out/pmc_knn}; N=${2:-1000000}
mkdir -p $out; export TMPDIR=/tmp
rocprofv3 -L > $out/counters_available.txt 2>&1
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python tools/knn_only.py $N 1 > $out/log_p$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn16_topk" in k:
            key = ("nprod1" if any(t in k for t in ("ELi1EE", "ELi1ELb", ", 1>", ", 1, true>", ", 1, false>")) else "nprod3", r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key] += 1
    for key in sorted(acc):
        print(f.split("/")[-3] if "/" in f else f, key[0], key[1], "%.6g" % (acc[key] / n[key]), "(per dispatch, %d dispatches)" % n[key])
PY
