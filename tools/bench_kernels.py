#!/usr/bin/env python3
"""Per-kernel timing of the hot-path convs at the 256x256 bs=8 shapes.
Times forward / data-gradient / weight-gradient of each layer with HIP events on torch's
current stream (the stream the kernels are launched on) and prints TFLOP/s."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--iters', type=int, default=10)
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    dev = torch.device('cuda:0')
    N = args.n
    layers = [  # name, H, Cin, Cout, k, stride, pad
        ('head conv 1024->1024 @8', 8, 1024, 1024, 3, 1, 1),
        ('mid conv 1024->1024 @16', 16, 1024, 1024, 3, 1, 1),
        ('up_0 conv 1024->512 @32', 32, 1024, 512, 3, 1, 1),
        ('up_1 conv 512->256 @64', 64, 512, 256, 3, 1, 1),
        ('up_2 conv 256->128 @128', 128, 256, 128, 3, 1, 1),
        ('up_3 conv_0 128->64 @256', 256, 128, 64, 3, 1, 1),
        ('up_3 conv_1 64->64 @256', 256, 64, 64, 3, 1, 1),
        ('gb 128->256 @256 (C=128)', 256, 128, 256, 3, 1, 1),
        ('gb 128->128 @256 (C=64)', 256, 128, 128, 3, 1, 1),
        ('gb 128->512 @128 (C=256)', 128, 128, 512, 3, 1, 1),
        ('gb 128->2048 @16 (C=1024)', 16, 128, 2048, 3, 1, 1),
        ('D m1 64->128 4x4s2 @129', 129, 64, 128, 4, 2, 2),
        ('D m3 256->512 4x4s1 @33', 33, 256, 512, 4, 1, 2),
    ]
    rows = []
    for name, H, cin, cout, k, s, p in layers:
        n = 2 * N if name.startswith('D') else N
        x = torch.randn(n, H, H, cin, device=dev).to(dt)
        w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
        Ho = (H + 2 * p - k) // s + 1
        gy = torch.randn(n, Ho, Ho, cout, device=dev).to(dt)
        wp = ops.pack_weight(w, dt, cin, False)
        wpt = ops.pack_weight(w, dt, cin, True)
        flops = 2.0 * n * Ho * Ho * cout * cin * k * k
        t_f = timeit(lambda: ops.conv2d_raw(x, wp, None, None, None, (Ho, Ho, cout), k, k, s, p), args.iters)
        t_d = timeit(lambda: ops.conv2d_raw(gy, wpt, None, None, None, (H, H, cin), k, k, s, p, True), args.iters)
        t_w = timeit(lambda: ops.conv2d_wgrad_raw(x, gy, k, k, s, p), args.iters)
        rows.append(dict(layer=name, gflop=flops / 1e9, fwd_ms=t_f, dgrad_ms=t_d, wgrad_ms=t_w,
                         fwd_tf=flops / t_f / 1e9, dgrad_tf=flops / t_d / 1e9, wgrad_tf=flops / t_w / 1e9))
        print('%-28s %8.1f GF  fwd %7.3f ms %7.1f TF | dgrad %7.3f ms %7.1f TF | wgrad %7.3f ms %7.1f TF' % (
            name, flops / 1e9, t_f, flops / t_f / 1e9, t_d, flops / t_d / 1e9, t_w, flops / t_w / 1e9), flush=True)
    # HBM-bound kernels at the largest block (C=128 @256^2)
    x = torch.randn(N, 256, 256, 128, device=dev).to(dt)
    gb = torch.randn(N, 256, 256, 256, device=dev).to(dt)
    style = torch.randn(N, 256, device=dev)
    esz = x.element_size()
    t = timeit(lambda: ops.in_stats(x), args.iters)
    print('in_stats C=128@256: %.3f ms  %.0f GB/s' % (t, x.numel() * esz / t / 1e6))
    st = ops.in_stats(x)
    t = timeit(lambda: ops.ModulateFn.apply(x, gb, style, st, True), args.iters)
    print('modulate_fwd C=128@256: %.3f ms  %.0f GB/s' % (t, (2 * x.numel() + gb.numel()) * esz / t / 1e6))
    t = timeit(lambda: ops.upsample2x(x[:, :128, :128].contiguous()), args.iters)
    print(json.dumps(dict(dtype=args.dtype, rows=rows)))


if __name__ == '__main__':
    main()
