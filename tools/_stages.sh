#!/bin/bash
export MELD_DEV=1   # (development switches are read only under MELD_DEV=1: meld_amd/_options.py)
# stage timers of one bench run under the given environment: bash tools/_stages.sh [VAR=val ...]
env "$@" python bench.py --steps 8 --warmup 3 --cpu-sample 0 --no-host-input --no-extra --stages 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.2f' % d['ms_per_step'], 'knn %.2f' % d['roofline']['ms'], {k: round(v*1e3,2) for k,v in d['stages'].items()})"
