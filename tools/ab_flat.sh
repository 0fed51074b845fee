for r in 1 2; do
for m in 0 48; do
S2E_WGRAD_FLAT=$m python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flat=$m', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
