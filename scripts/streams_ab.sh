for S in 3 4 5 3 4 5; do
  for K in 20; do
    python bench.py --steps $K --warmup 5 --streams $S --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=$S K=$K', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
  done
done
for S in 3 4; do python bench.py --steps 120 --warmup 5 --streams $S --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=$S K=120', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
