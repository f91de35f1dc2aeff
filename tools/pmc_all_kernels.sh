#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# Which unit bounds which kernel of a step: counters-only passes (one set per run, --kernel-trace only) over one bench step,
# reduced per kernel name to per-dispatch means and three ratios.  tools/pmc_all_kernels.sh <outdir> [cells]
out=${1:-gpurun_out/pmc_all}; N=${2:-1000000}
mkdir -p $out; export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python bench.py --cells $N --steps 2 --warmup 1 --no-extra --cpu-sample 0 --no-host-input > $out/log_p$i.txt 2>&1
  echo "pass $i ($set): rc=$?"
done <<SETS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
SETS
python - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
rows = []
for k in acc:
    m = {c: acc[k][c] / n[k][c] for c in acc[k]}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0          # cycles of the dispatch (the counter sums the 8 XCDs)
    if cyc <= 0: continue
    calls = n[k]["GRBM_GUI_ACTIVE"]
    rows.append((cyc * calls, k, calls, cyc, m))
rows.sort(reverse=True)
print("%-60s %5s %9s | %6s %6s %6s | %6s %7s %7s" % ("kernel", "calls", "cyc/call", "L1/cyc", "valu", "lds", "L2hit", "fabGB", "L1->L2"))
for tot, k, calls, cyc, m in rows[:40]:
    l1 = m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256.0 / cyc                      # L1 tag accesses per CU and cycle
    valu = 4.0 * m.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * cyc)                    # share of SIMD cycles issuing VALU (quad-cycle counter)
    lds = 4.0 * m.get("SQ_ACTIVE_INST_LDS", 0) / (1024.0 * cyc)
    hit = m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0))
    fab = (m.get("TCC_EA0_RDREQ_sum", 0) * 128 + m.get("TCC_EA0_WRREQ_sum", 0) * 64) / 1e9  # upper estimate (a read request is 64 or 128 B)
    l12 = (m.get("TCP_TCC_READ_REQ_sum", 0) + m.get("TCP_TCC_WRITE_REQ_sum", 0)) * 64 / 1e9
    print("%-60s %5d %9.0f | %6.2f %6.2f %6.2f | %6.2f %7.2f %7.2f" % (k, calls, cyc, l1, valu, lds, hit, fab, l12))
PY
