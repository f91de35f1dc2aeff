#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# PMC passes over the recurrence kernel (run on the GPU box): tools/pmc_spmm.sh <outdir> [N] [ablate-mask]
# One rocprofv3 run per counter set (counters + kernel trace only).  Prints per-dispatch averages per kernel.
out=${1:-gpurun_out/pmc_spmm}; N=${2:-1000000}; AB=${3:-0}
mkdir -p $out; export TMPDIR=/tmp
[ -f /tmp/g_$N.pt ] || python tools/save_graph.py $N /tmp/g_$N.pt
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  PT_MASK=$AB timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python tools/spmm_time.py /tmp/g_$N.pt 4 > $out/log_p$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_MISC
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "elementwise_kernel" in k and "MulFunctor" in k:
            k = "plain copy (torch)("
        if "pt_step" in k or k.startswith("plain copy"):
            short = k.split("(")[0].replace("void meld::", "")
            key = (short, r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key] += 1
    for key in sorted(acc):
        print("%-34s %-32s %.6g  (avg of %d dispatches)" % (key[0], key[1], acc[key] / n[key], n[key]))
PY
