#!/usr/bin/env python3
"""HBM bytes per launch of the conv kernel families from two rocprofv3 PMC passes.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events
    python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w profiles/r01/pmc

Counters are in KB; FETCH_SIZE is doubled (gfx950 counts a 128-B request as 64 B: MI355X_MICROARCH.md, HBM section).
Families follow seg2eye_amd.ops.LaunchProfiler (= s2e_conv2d_kernel_kind / s2e_conv2d_wgrad_kernel_kind): `conv_patch`,
`conv_igemm` (generic), `conv_small`, and the same three for the weight gradient; the per-family figure is total bytes /
number of C-ABI calls (= launches of the family's MAIN kernel; helper kernels -- split-K finish, partial-tile reduction --
add bytes, not launches)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = {  # kernel-name substring -> (family, counts as a launch of the family)
    'conv_patch_kernel': ('conv_patch', True),
    'conv_igemm_kernel': ('conv_igemm', True),
    'conv_finish_kernel': ('conv_igemm', False),          # (also finishes the patch kernel's channel-chunk splits)
    'fwd_cout1_kernel': ('conv_small', True), 'fwd_cin1_kernel': ('conv_small', True), 'dgrad_cout1_kernel': ('conv_small', True),
    'conv_wgrad_patch_kernel': ('conv_wgrad_patch', True),
    'wgrad_patch_reduce_kernel': ('conv_wgrad_patch', False),
    'conv_wgrad_c8_kernel': ('conv_wgrad_patch', True), 'wgrad_c8_reduce_kernel': ('conv_wgrad_patch', False),
    'conv_wgrad_kernel': ('conv_wgrad', True), 'conv_wgrad_glds_kernel': ('conv_wgrad', True),
    'wgrad_cout1_kernel': ('conv_wgrad_small', True), 'wgrad_cin1_kernel': ('conv_wgrad_small', True),
    'small_wgrad_reduce_kernel': ('conv_wgrad_small', False),
}


def classify(name):
    for key in sorted(FAMILIES, key=len, reverse=True):
        if key in name:
            return key, FAMILIES[key]
    return None, (None, False)


def read(dirname, counter):
    per_kernel = defaultdict(lambda: [0, 0.0])          # kernel key -> [dispatches, KB]
    files = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under %s' % dirname
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            key, _ = classify(row['Kernel_Name'])
            if key is None:
                continue
            e = per_kernel[key]
            e[0] += 1
            e[1] += float(row['Counter_Value'])
    return per_kernel


def main():
    fdir, wdir, outdir = sys.argv[1:4]
    fetch, write = read(fdir, 'FETCH_SIZE'), read(wdir, 'WRITE_SIZE')
    fams = defaultdict(lambda: {'launches': 0, 'fetch': 0.0, 'write': 0.0})
    rows = []
    for key in sorted(set(fetch) | set(write)):
        fam, counts = FAMILIES[key]
        n = fetch.get(key, [0, 0.0])[0] or write.get(key, [0, 0.0])[0]
        fb = fetch.get(key, [0, 0.0])[1] * 1024.0 * 2.0
        wb = write.get(key, [0, 0.0])[1] * 1024.0
        rows.append((key, fam, n, fb / max(n, 1), wb / max(n, 1)))
        fams[fam]['fetch'] += fb
        fams[fam]['write'] += wb
        if counts:
            fams[fam]['launches'] += n
    out = {'_how': __doc__.split('\n\n')[1].strip() + ' | counters KB -> bytes, FETCH_SIZE doubled (gfx950); per-family: '
           'total bytes of every kernel launched inside the C-ABI call / number of calls', 'kernels': {}}
    for fam, v in fams.items():
        n = max(v['launches'], 1)
        out['kernels'][fam] = {'launches': v['launches'], 'fetch_bytes_per_launch': v['fetch'] / n,
                               'write_bytes_per_launch': v['write'] / n, 'hbm_bytes_per_launch': (v['fetch'] + v['write']) / n}
    os.makedirs(outdir, exist_ok=True)
    json.dump(out, open(os.path.join(outdir, 'hbm_traffic.json'), 'w'), indent=1)
    with open(os.path.join(outdir, 'hbm_traffic_per_kernel.csv'), 'w') as f:
        f.write('kernel,family,dispatches,fetch_bytes_per_dispatch,write_bytes_per_dispatch\n')
        for r in rows:
            f.write('%s,%s,%d,%.0f,%.0f\n' % r)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
