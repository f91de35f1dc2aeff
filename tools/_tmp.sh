timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --stages 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('ms/step %.1f' % d['ms_per_step'], d['value'], {k: round(v*1e3,1) for k,v in d['stages'].items() if v > 0.5e-3})"
