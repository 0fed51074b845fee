#!/bin/bash
# per-step time of the kernels whose name matches a pattern, under the kernel trace of the bench's step:  bash tools/kt_kernels.sh "<regex>" [ENV=VALUE ...]
export TMPDIR=/tmp
PAT=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/ktk
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktk -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events > /dev/null 2>&1
f=$(ls /tmp/ktk/*/*kernel_stats.csv /tmp/ktk/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" "$PAT" "$*" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max(int(r['Calls']) for r in rows if 'adam_flat' in r['Name']) / 2.0
tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6
print('[%s] all kernels %.3f ms per step' % (sys.argv[3], tot))
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print('   %8.1f us/step  %5.1f x %7.1f us  %s' % (float(r['TotalDurationNs']) / steps / 1e3, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
