#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# PMC passes over refine_kernel (run on the GPU box): tools/pmc_refine.sh <outdir> [N]   -- counters only, one set per run
out=${1:-gpurun_out/pmc_refine}; N=${2:-1000000}
mkdir -p $out; export TMPDIR=/tmp
rocprofv3 -L 2>&1 | grep -o "T[AC][A-Z]*_[A-Za-z_0-9]*" | sort -u > $out/ta_tc_counter_names.txt
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python tools/knn_only.py $N 1 > $out/log_p$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done <<SETS
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "refine_kernel" in k and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) > 1000000:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for key in sorted(acc):
        print(key, "%.6g" % (acc[key] / n[key]), "(per dispatch, %d dispatches)" % n[key])
PY
