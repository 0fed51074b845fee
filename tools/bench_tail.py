#!/usr/bin/env python3
"""The generic-kernel tail of the G+D step: forward, data-gradient and weight-gradient of every layer
that the patch-resident kernels do not take, at the bench shapes (256x256, batch 8: netD sees 16 images, netE 32),
timed with HIP events on the launch stream and checked against torch's own convolution in fp32.

    S2E_CONV_STREAM=0 python tools/bench_tail.py      # the round-1 implicit GEMM + split-K finish
    S2E_CONV_STREAM=1 python tools/bench_tail.py      # the persistent stream-K kernel (default)
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops  # noqa: E402
from seg2eye_amd import _lib as L  # noqa: E402

# name, N, H (= W) of the forward input, Cin, Cout, k, stride, pad, which passes the step runs in a generic kernel
LAYERS = [
    ('D0 m0 8->64 4x4s2 @256', 16, 256, 8, 64, 4, 2, 2, 'FDW'),
    ('D0 m1 64->128 4x4s2 @129', 16, 129, 64, 128, 4, 2, 2, 'FDW'),
    ('D0 m2 128->256 4x4s2 @65', 16, 65, 128, 256, 4, 2, 2, 'FDW'),
    ('D0 m3 256->512 4x4s1 @33', 16, 33, 256, 512, 4, 1, 2, 'W'),
    ('D1 m0 8->64 4x4s2 @128', 16, 128, 8, 64, 4, 2, 2, 'FDW'),
    ('D1 m1 64->128 4x4s2 @65', 16, 65, 64, 128, 4, 2, 2, 'FDW'),
    ('D1 m2 128->256 4x4s2 @33', 16, 33, 128, 256, 4, 2, 2, 'FDW'),
    ('D1 m3 256->512 4x4s1 @17', 16, 17, 256, 512, 4, 1, 2, 'FDW'),
    ('E l1 64->128 3x3s2 @128', 32, 128, 64, 128, 3, 2, 1, 'FDW'),
    ('E l2 128->256 3x3s2 @64', 32, 64, 128, 256, 3, 2, 1, 'FDW'),
    ('E l3 256->512 3x3s2 @32', 32, 32, 256, 512, 3, 2, 1, 'FDW'),
    ('E l4 512->512 3x3s2 @16', 32, 16, 512, 512, 3, 2, 1, 'FDW'),
    ('E l5 512->512 3x3s2 @8', 32, 8, 512, 512, 3, 2, 1, 'FDW'),
    ('G head 1024->1024 3x3 @8', 8, 8, 1024, 1024, 3, 1, 1, 'FDW'),
    ('G gb 128->2048 3x3 @16', 8, 16, 128, 2048, 3, 1, 1, 'D'),
    ('G gb 128->2048 3x3 @8', 8, 8, 128, 2048, 3, 1, 1, 'DW'),
    ('G s 128->64 1x1 @256', 8, 256, 128, 64, 1, 1, 0, 'FDW'),
    ('G s 256->128 1x1 @128', 8, 128, 256, 128, 1, 1, 0, 'FDW'),
    ('G s 512->256 1x1 @64', 8, 64, 512, 256, 1, 1, 0, 'FDW'),
    ('G s 1024->512 1x1 @32', 8, 32, 1024, 512, 1, 1, 0, 'FDW'),
    ('x odd 96->192 3x3 @128', 8, 128, 96, 192, 3, 1, 1, 'FDW'),
]


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def relerr(a, ref):
    return float((a.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dt = torch.bfloat16
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    rows, tot = [], dict(F=0.0, D=0.0, W=0.0)
    worst = 0.0
    for name, n, H, cin, cout, k, s, p, passes in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn(n, H, H, cin, device=dev).to(dt)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, device=dev)
        Ho = (H + 2 * p - k) // s + 1
        gy = torch.randn(n, Ho, Ho, cout, device=dev).to(dt)
        wp = ops.pack_weight(w, dt, cin, False)
        wpt = ops.pack_weight(w, dt, cin, True)
        flops = 2.0 * n * Ho * Ho * cout * cin * k * k
        wq = w.to(dt).float()
        r = dict(layer=name, gflop=flops / 1e9)
        line = '%-28s %7.1f GF' % (name, flops / 1e9)
        if 'F' in passes:
            y = ops.conv2d_raw(x, wp, bias, None, None, (Ho, Ho, cout), k, k, s, p)
            if not args.no_check:
                ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, bias, stride=s, padding=p).permute(0, 2, 3, 1)
                e = relerr(y, ref); worst = max(worst, e); r['fwd_err'] = e
            t = timeit(lambda: ops.conv2d_raw(x, wp, bias, None, None, (Ho, Ho, cout), k, k, s, p), args.iters)
            r['fwd_ms'] = t; tot['F'] += t
            line += ' | F %6.1f us %6.1f TF' % (t * 1e3, flops / t / 1e9)
        if 'D' in passes:
            gx = ops.conv2d_raw(gy, wpt, None, None, x, (H, H, cin), k, k, s, p, True, L.ACT_NONE, L.ACT_NONE, L.AUX_RELU_MASK)
            if not args.no_check:
                ref = F.conv_transpose2d(gy.float().permute(0, 3, 1, 2), wq, stride=s, padding=p,
                                         output_padding=H - ((Ho - 1) * s - 2 * p + k)).permute(0, 2, 3, 1)
                ref = ref * (x.float() > 0)
                e = relerr(gx, ref); worst = max(worst, e); r['dgrad_err'] = e
            t = timeit(lambda: ops.conv2d_raw(gy, wpt, None, None, None, (H, H, cin), k, k, s, p, True), args.iters)
            r['dgrad_ms'] = t; tot['D'] += t
            line += ' | D %6.1f us %6.1f TF' % (t * 1e3, flops / t / 1e9)
        if 'W' in passes:
            dw, db = ops.conv2d_wgrad_raw(x, gy, k, k, s, p, L.ACT_NONE, True)
            if not args.no_check:
                xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
                wr = wq.clone().requires_grad_(True)
                out = F.conv2d(xr, wr, None, stride=s, padding=p)
                out.backward(gy.float().permute(0, 3, 1, 2))
                ref = wr.grad.permute(0, 2, 3, 1).reshape(cout, -1)           # (Cout, KH*KW*Cin) packed order
                e = relerr(dw, ref); worst = max(worst, e); r['wgrad_err'] = e
                eb = relerr(db, gy.float().sum((0, 1, 2))); worst = max(worst, eb); r['dbias_err'] = eb
            t = timeit(lambda: ops.conv2d_wgrad_raw(x, gy, k, k, s, p, L.ACT_NONE, True), args.iters)
            r['wgrad_ms'] = t; tot['W'] += t
            line += ' | W %6.1f us %6.1f TF' % (t * 1e3, flops / t / 1e9)
        errs = [r.get(k_) for k_ in ('fwd_err', 'dgrad_err', 'wgrad_err', 'dbias_err') if r.get(k_) is not None]
        if errs:
            line += ' | err %.1e' % max(errs)
        print(line, flush=True)
        rows.append(r)
    print('total over the list: F %.3f ms, D %.3f ms, W %.3f ms; worst relative error %.2e' % (tot['F'], tot['D'], tot['W'], worst))
    print(json.dumps(dict(stream=os.environ.get('S2E_CONV_STREAM', '1'), rows=rows, total=tot, worst_err=worst)))
    if not args.no_check and worst > 2e-2:
        sys.exit('FAILED: relative error %.3e' % worst)


if __name__ == '__main__':
    main()
