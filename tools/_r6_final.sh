#!/bin/bash
# the closing evidence of round 6 at the round's last product commit: bash tools/_r6_final.sh <commit>  (on the GPU box)
export MELD_DEV=1
commit=${1:-unknown}; out=gpurun_out/final_r06; mkdir -p $out; export TMPDIR=/tmp
stamp() { echo "# commit $commit, $(date -u +%Y-%m-%dT%H:%MZ)"; }
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $out/gpu_tests.txt; cat $out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
env -u MELD_DEV python bench.py 2>$out/bench.err > $out/bench_1M.json; python - $out/bench_1M.json $commit <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); d["commit"] = sys.argv[2]
open(sys.argv[1], "w").write(json.dumps(d) + "\n")
print(d["value"], d["ms_per_step"], d.get("value_host_input"), d["roofline"]["ms"], d["roofline"]["frac"], (d.get("roofline_filter") or {}).get("ms"), d["roofline_cheby"]["frac"], d["roofline_cheby_c3"]["frac"], d.get("roofline_d100"))
PY
{ stamp; echo "# stage timers of a step (bench.py --stages, 1M x 50)"; python bench.py --cpu-sample 0 --no-host-input --no-extra --stages 2>/dev/null | python tools/_benchline.py
  echo "# other sizes (bench.py --cells N --steps 5 --no-extra --cpu-sample 0 --no-host-input): ms per step"
  for n in 100000 200000 300000 400000 500000 2000000 4000000; do echo -n "N=$n: "; python bench.py --cells $n --steps 5 --no-extra --cpu-sample 0 --no-host-input 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  %.2f M cells/s' % (d['ms_per_step'], d['value']/1e6))"; done
  echo "# 1M x 100 (d = 100)"; python bench.py --cells 1000000 --dims 100 --steps 3 --no-extra --cpu-sample 0 --no-host-input 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  %.2f M cells/s' % (d['ms_per_step'], d['value']/1e6))"
  } > $out/stages_and_sizes.txt
{ stamp; echo "# per-rank compute of the sharded driver on ONE GPU (stand-in collectives): tools/shard_emulate.py"
  for g in 2 4 8; do python tools/shard_emulate.py 1000000 $g 1 2>&1 | grep "^rank\|^collectives\|^lmax" | tail -3; done
  for g in 8; do for r in 1 0; do echo "## world $g, RCCL=$r"; RCCL=$r python tools/shard_emulate.py 1000000 $g 0 2>&1 | grep "^rank\|^collectives\|^lmax" | tail -3; done; done; } > $out/shard_emulation.txt
cat $out/stages_and_sizes.txt | cut -c1-400
