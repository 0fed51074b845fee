#!/bin/bash
# kernel trace of the step's generic / flat weight-gradient launches for several S2E_WGRAD_FLAT masks:  bash tools/kt_wgrad.sh "0 16 48 56"
export TMPDIR=/tmp
for m in ${1:-0 56}; do
  export S2E_WGRAD_FLAT=$m
  rm -rf /tmp/kt$m
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$m -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events > /dev/null 2>&1
  f=$(ls /tmp/kt$m/*/*kernel_stats.csv /tmp/kt$m/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" $m <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sel = [r for r in rows if any(k in r['Name'] for k in ('conv_wgrad_flat_kernel', 'conv_wgrad_multi_kernel', 'conv_wgrad_reduce_multi_kernel', 'conv_c8s2_wgrad'))]
calls = max(int(r['Calls']) for r in rows if 'adam_flat_kernel' in r['Name']) / 2.0 + 2      # step bodies profiled (two eager ones before the capture)
tot = sum(float(r['TotalDurationNs']) for r in sel) / calls / 1e3
print('S2E_WGRAD_FLAT=%s: %.1f us per step in' % (sys.argv[2], tot), ', '.join('%s %.1f x %d' % (r['Name'].split('(')[-2].split(':')[-1][:28] if '(' in r['Name'] else r['Name'][:28], float(r['AverageNs']) / 1e3, int(r['Calls'])) for r in sel))
PY
done
