#!/bin/bash
# round 5's last code (a git worktree of 1a67039 under _r05, its own build) against this tree on ONE box, ABAB:  bash tools/ab_r05.sh
for r in 1 2; do
for d in _r05 .; do
(cd $d && python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['value'],1), round(d['ms_per_step'],3), d.get('single_batch',{}).get('value'))")
done; done
