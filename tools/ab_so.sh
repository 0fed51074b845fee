#!/bin/bash
# same-box A/B of two builds of the library (_ab/old.so, _ab/new.so: copied over the in-tree one in turn; ABAB):  bash tools/ab_so.sh ["<kernel regex>"]
cp seg2eye_amd/lib/libseg2eye_hip.so /tmp/keep.so
for r in 1 2; do
for v in old new; do
cp _ab/$v.so seg2eye_amd/lib/libseg2eye_hip.so
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
if [ -n "$1" ]; then for v in old new; do cp _ab/$v.so seg2eye_amd/lib/libseg2eye_hip.so; bash tools/kt_kernels.sh "$1" AB=$v; done; fi
cp /tmp/keep.so seg2eye_amd/lib/libseg2eye_hip.so
