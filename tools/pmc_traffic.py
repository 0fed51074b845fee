#!/usr/bin/env python3
"""HBM bytes per launch of every kernel family of the G+D step from two rocprofv3 PMC passes.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-kernel-events
    python3 tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w profiles/r06/pmc [steps-profiled]

Counters are in KB; FETCH_SIZE is doubled (gfx950 counts a 128-B request as 64 B: MI355X_MICROARCH.md, HBM section).
Families are seg2eye_amd.ops.LaunchProfiler's -- one per C-ABI entry point (the conv entry points split by
s2e_conv2d_kernel_kind / s2e_conv2d_wgrad_kernel_kind) -- so the per-launch figure divides the bytes of EVERY kernel a call
launches (split-K finish, partial-tile reductions, the three kernels of s2e_modulate_bwd ...) by the number of calls (= the
dispatches of the family's MAIN kernel).  The output is stamped with the commit it was measured at."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

FAMILIES = {  # kernel-name substring -> (family, counts as a launch of the family)
    'conv_patch_kernel': ('conv_patch', True),
    'conv_duo_kernel': ('conv_patch', True),            # (round 4: the same s2e_conv2d / s2e_spade_conv_modulate launches, two workgroups per CU)
    'conv_plane_kernel': ('conv_plane', True),            # (round 6: netE's stride-2 3x3 layers and the 1x1 shortcuts, csrc/conv_plane.hip)
    'conv_igemm_kernel': ('conv_igemm', True),
    'conv_stream_kernel': ('conv_igemm', True), 'conv_stream_fixup_kernel': ('conv_igemm', False),   # (same profiler family: s2e_conv2d's generic shapes)
    'conv_finish_kernel': ('conv_igemm', False),          # (also finishes the patch kernel's channel-chunk splits)
    'fwd_cout1_kernel': ('conv_small', True), 'fwd_cin1_kernel': ('conv_small', True), 'dgrad_cout1_kernel': ('conv_small', True),
    'conv_wgrad_patch_kernel': ('conv_wgrad_patch', True),
    # (round 5: the queued patch-resident weight gradients of a backward as ONE launch + a fix-up: csrc/conv_wgrad_batch.hip)
    'conv_wgrad_batch_kernel': ('conv_wgrad_patch', True), 'wgrad_batch_fixup_kernel': ('conv_wgrad_patch', False),
    'wgrad_patch_reduce_kernel': ('conv_wgrad_patch', False),
    'conv_wgrad_c8_kernel': ('conv_wgrad_patch', True), 'wgrad_c8_reduce_kernel': ('conv_wgrad_patch', False),
    'conv_wgrad_kernel': ('conv_wgrad', True), 'conv_wgrad_glds_kernel': ('conv_wgrad', True),
    'wgrad_cout1_kernel': ('conv_wgrad_small', True), 'wgrad_cin1_kernel': ('conv_wgrad_small', True),
    'small_wgrad_reduce_kernel': ('conv_wgrad_small', False),
    # HBM-bound families
    'in_stats_partial_kernel': ('in_stats', True), 'in_stats_finalize_kernel': ('in_stats', False),
    'in_stats_from_partials_kernel': ('in_stats', True),   # (round 4: the fold of partial sums a conv's epilogue wrote)
    # small maps, one launch per call: in_small_kernel<T, 0, G> = statistics only (s2e_in_stats), <T, 1, G> = the whole plain
    # InstanceNorm forward (s2e_instance_norm_fwd, family modulate_fwd).  rocprof prints the template arguments either as
    # '<..., 0, ...>' or mangled 'Li0E': both spellings are listed
    'in_small_kernelIDF16bLi0E': ('in_stats', True), 'in_small_kernelIfLi0E': ('in_stats', True),
    'in_small_kernelIDF16bLi1E': ('modulate_fwd', True), 'in_small_kernelIfLi1E': ('modulate_fwd', True),
    'in_small_kernel<': ('modulate_fwd', True),
    'in_small_bwd_kernel': ('modulate_bwd', True), 'spade_small_bwd_kernel': ('modulate_bwd', True),
    'modulate_fwd_kernel': ('modulate_fwd', True),
    'spade_modulate_uniform_kernel': ('modulate_fwd', True),   # label-uniform rectangles of a label-sparse SPADE forward
    'label_rect_classify_kernel': ('label_rects', True), 'label_rect_compact_kernel': ('label_rects', False),
    'spade_class_table_kernel': ('class_table', True),
    # label-sparse SPADE backward (round 4): nine shifted sums per uniform rectangle, then the closed-form gradients at the flush
    'spade_uniform_sums_kernel': ('spade_uniform_bwd', True), 'spade_uni_gemv_kernel': ('spade_uniform_bwd', True),
    'spade_uni_apply_kernel': ('spade_uniform_bwd', False), 'rect_lists_bwd_kernel': ('label_rects', False),
    'modulate_bwd_reduce_kernel': ('modulate_bwd', True), 'modulate_bwd_coef_kernel': ('modulate_bwd', False),
    'modulate_bwd_apply_kernel': ('modulate_bwd', False),
    'label_conv3x3_kernel': ('label_conv', True), 'label_conv3x3_batch_kernel': ('label_conv', True),
    'spade_class_table_batch_kernel': ('class_table', True),
    'modulate_bwd_apply_quad_kernel': ('modulate_bwd', False),
    'conv_wgrad_c8_batch_kernel': ('conv_wgrad_patch', False), 'wgrad_c8_batch_reduce_kernel': ('conv_wgrad_patch', False),
    'fc_head_fwd_kernel': ('fc_head', True), 'fc_head_bwd_kernel': ('fc_head', True),
    'lrelu_bwd_kernel': ('activation_bwd', True), 'tanh_bwd_kernel': ('activation_bwd', True),
    'pack_tr_cl_kernel': ('weight_pack', False), 'pack_fwd_cl_kernel': ('weight_pack', False), 'colsum_scalar_kernel': ('colsum', True), 'colsum_kernel': ('colsum', True),
    'adam_tick_kernel': ('adam', False),
    # stock torch / runtime (fills, copies, loss-scalar arithmetic)
    'at::native': ('torch', True), '__amd_rocclr': ('torch', True),
    'adam_flat_kernel': ('adam', True),
    'pack_batch_kernel': ('weight_pack', True),
    'grad_unpack_batch_kernel': ('weight_grad_relayout', True), 'grad_dot_batch_kernel': ('weight_grad_relayout', False),
    # channels-last masters (round 3): the in-place spectral-norm chain rule is the same profiler family
    'sn_grad_inplace_apply_kernel': ('weight_grad_relayout', True), 'sn_grad_inplace_dot_kernel': ('weight_grad_relayout', False),
    'style_fc_fwd_kernel': ('style_fc', True), 'style_fc_bwd_kernel': ('style_fc', True), 'style_fc_dw_fold_kernel': ('style_fc', False),
    'sn_gemvT_chain_kernel': ('spectral_norm', True), 'sn_gemv_chain_kernel': ('spectral_norm', False), 'sn_finalize_chain_kernel': ('spectral_norm', False),
    'conv_wgrad_reduce_kernel': ('conv_wgrad', False),
    # (round 6: every generic weight gradient of a backward as one multi-job launch + one reduction launch)
    'conv_wgrad_multi_kernel': ('conv_wgrad', True), 'conv_wgrad_reduce_multi_kernel': ('conv_wgrad', False),
    # (round 6, after milestone b: the PatchGAN's 4x4 weight gradients in the flat-slab kernel -- the D step's call has no generic job left --
    #  and its 8-channel first layer in conv_c8.hip)
    'conv_wgrad_flat_kernel': ('conv_wgrad', True), 'conv_c8s2_wgrad_kernel': ('conv_wgrad', False), 'conv_c8s2_wgrad_reduce_kernel': ('conv_wgrad', False),
    'conv_c8s2_fwd_kernel': ('conv_small', True), 'conv_c8s2_dgrad_kernel': ('conv_small', True),
    'sn_gemvT_kernel': ('spectral_norm', True), 'sn_gemv_kernel': ('spectral_norm', False),     # launches = power ITERATIONS
    'sn_norm_v_kernel': ('spectral_norm', False), 'sn_finalize_kernel': ('spectral_norm', False),
    'upsample2x_fwd_kernel': ('resample', True), 'upsample2x_bwd_kernel': ('resample', True),
    'avgpool_fwd_kernel': ('resample', True), 'avgpool_bwd_kernel': ('resample', True),
    'loss_reduce_kernel': ('loss', True), 'loss_grad_kernel': ('loss', True),
    'onehot_nhwc_kernel': ('onehot', True), 'onehot_nhwc8_kernel': ('onehot', True),
    # round 5, after milestone a: the degenerate-channel convs on the matrix cores, the two-launch encoder head, vector pools
    'tap_gemm_kernel': ('conv_small', True), 'fwd_cout1_mfma_kernel': ('conv_small', True),
    'fc_head_part_kernel': ('fc_head', True), 'fc_head_fin_kernel': ('fc_head', False),
    'avgpool_fwd_vec_kernel': ('resample', True), 'avgpool_bwd_vec_kernel': ('resample', True),
    'pack_tr_kernel': ('weight_pack', False), 'pack_fwd_kernel': ('weight_pack', False),
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


def source_digest():
    """sha1 over the kernel sources and the host package (what a profile is a profile OF): bench.py recomputes it on the box it runs on
    -- there is no .git there -- and marks a quoted profile `rocprof_stale` when it differs."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for pat in ('seg2eye_amd/csrc/*', 'seg2eye_amd/*.py', 'seg2eye_amd/*/*.py', 'include/*.h'):
        for f in sorted(glob.glob(os.path.join(root, pat))):
            if os.path.isfile(f):
                h.update(os.path.relpath(f, root).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def git_head():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        head = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(['git', '-C', root, 'status', '--porcelain', '--', 'seg2eye_amd', 'bench.py'], capture_output=True, text=True).stdout.strip()
        return (head or os.environ.get('S2E_GIT_HEAD', 'unknown')) + ('+dirty' if dirty else '')
    except OSError:
        return os.environ.get('S2E_GIT_HEAD', 'unknown')


def main():
    fdir, wdir, outdir = sys.argv[1:4]
    fetch, write = read(fdir, 'FETCH_SIZE'), read(wdir, 'WRITE_SIZE')
    # steps profiled = Adam launches / 2 (one per optimizer per G+D step): bench.py runs warm-up + timed + per-step-timed (+ the
    # extras') steps, so a count passed by hand goes stale (3 was written for a command that runs 5)
    adam = [v[0] for k, v in fetch.items() if 'adam_flat' in k] or [v[0] for k, v in write.items() if 'adam_flat' in k]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else (max(1, adam[0] // 2) if adam else 1)
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
    # the head is passed in by the caller when the tool runs on the GPU box (no .git there)
    out = {'git_head': os.environ.get('S2E_GIT_HEAD') or git_head(), 'source_digest': source_digest(), 'steps_profiled': steps,
           '_how': __doc__.split('\n\n')[1].strip() + ' | counters KB -> bytes, FETCH_SIZE doubled (gfx950); per-family: '
           'total bytes of every kernel launched inside the C-ABI call / number of calls', 'kernels': {}}
    for fam, v in fams.items():
        n = max(v['launches'], 1)
        out['kernels'][fam] = {'launches': v['launches'], 'fetch_bytes_per_launch': v['fetch'] / n,
                               'write_bytes_per_launch': v['write'] / n, 'hbm_bytes_per_launch': (v['fetch'] + v['write']) / n,
                               'hbm_bytes_per_step': (v['fetch'] + v['write']) / steps}
    os.makedirs(outdir, exist_ok=True)
    json.dump(out, open(os.path.join(outdir, 'hbm_traffic.json'), 'w'), indent=1)
    with open(os.path.join(outdir, 'hbm_traffic_per_kernel.csv'), 'w') as f:
        f.write('kernel,family,dispatches,fetch_bytes_per_dispatch,write_bytes_per_dispatch\n')
        for r in rows:
            f.write('%s,%s,%d,%.0f,%.0f\n' % r)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
