python bench.py --force-sharded --steps 2 --warmup 1 --no-extra --cpu-sample 0 --no-host-input 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('collective_bytes_per_step'))
print(json.dumps(d.get('per_rank'))[:900])"
tail -3 /tmp/err.txt
