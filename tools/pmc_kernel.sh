#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# L2 / fabric counters of every kernel whose name matches a pattern, over one search stage (run on the GPU box):
#   [CMD="python tools/time_wide.py"] bash tools/pmc_kernel.sh <pattern> [N]
pat=${1:-refine_kernel}; N=${2:-1000000}; out=/tmp/pmc_kernel; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- ${CMD:-python tools/knn_only.py $N 1} > $out/log_p$i.txt 2>&1
done
python - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-36s total %.6g over %d dispatches" % (k, acc[k], n[k]))
PY
