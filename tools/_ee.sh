#!/bin/bash
# early-exit search: timings (EE on / off, rotation on / off), then a correctness subset
python tools/time_rotate.py 2>&1 | tail -3
echo "== default"; python tools/knn_only.py 1000000 3 2>&1 | tail -3
echo "== no rotation"; MELD_KNN_ROTATE=0 python tools/knn_only.py 1000000 3 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-host-input --no-extra --stages 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'knn ms', d['roofline']['ms'], {k: round(v*1e3,2) for k,v in d['stages'].items()})"
python bench.py --cells 500000 --steps 10 --warmup 3 --cpu-sample 0 --no-host-input --no-extra --stages 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('500k ms_per_step', d['ms_per_step'], 'knn ms', d['roofline']['ms'], {k: round(v*1e3,2) for k,v in d['stages'].items()})"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -5
