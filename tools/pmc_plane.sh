#!/bin/bash
# SQ + cache counters of one conv shape:  bash tools/pmc_plane.sh <out file> <bench_plane_one.py args...>
OUT=$1; shift
R=$(pwd); export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pp1 /tmp/pp2
(cd "$R" && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pp1 -o a -- python3 tools/bench_plane_one.py "$@" > /dev/null 2>&1)
(cd "$R" && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pp2 -o b -- python3 tools/bench_plane_one.py "$@" > /dev/null 2>&1)
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set); dur = defaultdict(float)
for d in ('/tmp/pp1', '/tmp/pp2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            if 'conv_' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if (d, r['Dispatch_Id']) not in n[k]:
                n[k].add((d, r['Dispatch_Id']))
                if d == '/tmp/pp1': dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
out = []
for k, c in acc.items():
    nd = len([1 for x in n[k] if x[0] == '/tmp/pp1'])
    wc = max(c['SQ_WAVE_CYCLES'], 1); gui = c['GRBM_GUI_ACTIVE'] / 8
    out.append('%s: %d disp, %.1f us avg, parked %.1f%% stalled %.1f%% issuing %.1f%%, mfma util %.1f%%, clk %.2f GHz, lds conflicts %.3g' % (
        k, nd, dur[k] / max(nd, 1) / 1e3, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc,
        100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui) if gui else 0, gui / dur[k] if dur[k] else 0, c['SQ_LDS_BANK_CONFLICT']))
    out.append('   per dispatch: ' + ', '.join('%s %.3g' % (kk, v / max(nd, 1)) for kk, v in sorted(c.items())))
open(sys.argv[1], 'a').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
