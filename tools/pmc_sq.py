#!/usr/bin/env python3
"""Wave-state split and matrix-pipe utilisation of the patch-resident kernels from one rocprofv3 SQ pass over bench.py:

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
        SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_sq -o sq -- python3 bench.py \
        --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events
    python3 tools/pmc_sq.py /tmp/pmc_sq profiles/r06/pmc/sq_counters_patch_kernels.txt

Per kernel (all dispatches of the run summed): SQ_WAIT_ANY (parked in s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue-stalled)
and SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (the three are disjoint, MI355X_MICROARCH.md); matrix-pipe utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) -- busy SIMD-cycles over the SIMD-cycles the dispatch lasted
(GRBM_GUI_ACTIVE sums the 8 XCDs); effective clock = GRBM_GUI_ACTIVE / 8 / duration from the kernel trace."""
import csv
import glob
import os
import sys
from collections import defaultdict

WANT = ('conv_plane_kernel', 'conv_wgrad_multi_kernel', 'conv_duo_kernel', 'conv_patch_kernel', 'conv_wgrad_batch_kernel', 'conv_wgrad_patch_kernel', 'conv_igemm_kernel', 'conv_stream_kernel', 'conv_wgrad_kernel', 'conv_wgrad_stream_kernel', 'conv_wgrad_flat_kernel', 'conv_c8s2_fwd_kernel', 'conv_c8s2_dgrad_kernel', 'conv_c8s2_wgrad_kernel', 'label_conv3x3_batch_kernel', 'spade_modulate_uniform_kernel', 'modulate_bwd_reduce_kernel', 'in_small_bwd_kernel')


def main():
    d, outp = sys.argv[1:3]
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    assert files, 'no counter_collection.csv under %s' % d
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            key = next((w for w in WANT if w in name), None)
            if key is None:
                continue
            tmpl = name[name.find('<'):name.find('>') + 1] if '<' in name else ''
            if not tmpl and 'IDF16b' in name:
                tmpl = name[name.find('I'):name.find('EEv') + 1]
            k = key + ' ' + tmpl[:48]
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            did = row.get('Dispatch_Id')
            if did not in disp[k]:
                disp[k].add(did)
                if row.get('Start_Timestamp') and row.get('End_Timestamp'):
                    dur[k] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
    lines = [__doc__.strip().split('\n\n')[0], '',
             '%-72s %6s %9s %7s %7s %7s %9s %8s %9s' % ('kernel', 'disp', 'ms total', 'parked', 'stalled', 'issuing', 'mfma util', 'clk GHz', 'lds confl')]
    for k in sorted(acc, key=lambda k: -dur[k]):
        c = acc[k]
        wc = max(c['SQ_WAVE_CYCLES'], 1.0)
        gui = c['GRBM_GUI_ACTIVE'] / 8.0
        util = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * gui) if gui else float('nan')
        clk = gui / dur[k] if dur[k] else float('nan')
        lines.append('%-72s %6d %9.2f %6.1f%% %6.1f%% %6.1f%% %8.1f%% %8.2f %9.3g' % (
            k, len(disp[k]), dur[k] / 1e6, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc,
            100 * util, clk, c['SQ_LDS_BANK_CONFLICT']))
    lines.append('')
    lines.append('git head: %s' % os.environ.get('S2E_GIT_HEAD', 'unknown'))
    open(outp, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
