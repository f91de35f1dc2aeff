timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --stages 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('ms/step %.1f' % d['ms_per_step'], d['value'], 'frac %.3f' % d['roofline']['frac'], {k: round(v*1e3,1) for k,v in d['stages'].items() if v > 0.5e-3})"
for cfg in "1000000 3" "1000000 100" "1000000 20" "200000 50"; do set -- $cfg; python bench.py --cells $1 --dims $2 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print(d['config']['workload'][:27], 'ms/step %.1f' % d['ms_per_step'], 'computed %.3f' % d['roofline']['blocks_computed_frac'], 'knn ms %.1f' % d['roofline']['ms'])"; done
