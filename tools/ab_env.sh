#!/bin/bash
# same-box A/B of one environment switch:  bash tools/ab_env.sh S2E_CONV_C8 0 1   (ABAB, graph replay, 50 steps each)
V=$1; A=$2; B=$3
for r in 1 2; do
for m in $A $B; do
env $V=$m python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$m', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
