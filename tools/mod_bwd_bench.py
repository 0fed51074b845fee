#!/usr/bin/env python3
"""Micro-benchmark of the SPADE+Style modulation backward (s2e_modulate_bwd_staged) and the plain InstanceNorm backward at the
step's large shapes: HIP-event time per call and algorithmic GB/s (DESIGN 3.5's access counts).  Run it under
`rocprofv3 --kernel-trace --stats` to split the passes (reduce / coef / apply).

    python tools/mod_bwd_bench.py [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import _lib as L                                          # noqa: E402
from seg2eye_amd import ops                                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    lib = L.lib()
    p = lambda t: t.data_ptr() if t is not None else None
    st = torch.cuda.current_stream().cuda_stream
    # (N, H, W, C, x at half resolution + quad dx)
    shapes = [(8, 256, 256, 128, True), (8, 256, 256, 128, False), (8, 256, 256, 64, False), (8, 128, 128, 256, True), (8, 128, 128, 256, False),
              (8, 128, 128, 128, False), (8, 64, 64, 512, True), (8, 64, 64, 512, False)]
    for (n, h, w, c, half) in shapes:
        torch.manual_seed(1)
        g = torch.randn(n, h, w, c, device=dev).bfloat16()
        x = torch.randn(n, h // 2, w // 2, c, device=dev).bfloat16() if half else torch.randn(n, h, w, c, device=dev).bfloat16()
        gamma = (0.1 * torch.randn(n, h, w, c, device=dev)).bfloat16()
        fout = torch.randn(n, h, w, c, device=dev).bfloat16()
        stats = torch.stack([torch.zeros(n, c, device=dev), torch.ones(n, c, device=dev)], -1).contiguous()
        style = (0.1 * torch.randn(n, 2 * c, device=dev)).contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(n, h, w, 2 * c, device=dev, dtype=torch.bfloat16)
        dstyle = torch.zeros(n, 2 * c, device=dev)
        ws = torch.empty(lib.s2e_modulate_bwd_workspace_bytes(L.S2E_BF16, n, h * w, c) // 8, dtype=torch.float64, device=dev)

        def call():
            L.check(lib.s2e_modulate_bwd_staged(L.S2E_BF16, L.NORM_SPADE_STYLE, p(g), p(x), p(gamma), p(fout), p(stats), p(style), p(dx), p(dgb),
                                                p(dstyle), p(ws), n, h * w, c, 1, 2 * c, 0, 0.0, w if half else 0, 1 if half else 0, st), 'bwd')
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        nb = (6.75 if half else 9.0) * n * h * w * c * 2
        print(f'spade_bwd n{n} {h}x{w} c{c} {"half-res x, quad dx" if half else "full-res x":20s} {us:8.1f} us  {nb / us / 1e3:7.0f} GB/s algorithmic', flush=True)
    for (n, h, w, c) in [(16, 129, 129, 128), (16, 65, 65, 256), (32, 128, 128, 64), (32, 64, 64, 128)]:
        g = torch.randn(n, h, w, c, device=dev).bfloat16()
        x = torch.randn(n, h, w, c, device=dev).bfloat16()
        stats = torch.stack([torch.zeros(n, c, device=dev), torch.ones(n, c, device=dev)], -1).contiguous()
        dx = torch.empty_like(x)
        ws = torch.empty(lib.s2e_modulate_bwd_workspace_bytes(L.S2E_BF16, n, h * w, c) // 8, dtype=torch.float64, device=dev)

        def call():
            L.check(lib.s2e_instance_norm_bwd(L.S2E_BF16, p(g), p(x), p(stats), p(dx), p(ws), n, h * w, c, 1, st), 'in_bwd')
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        nb = 5.0 * n * h * w * c * 2
        print(f'in_bwd    n{n} {h}x{w} c{c} {"":20s} {us:8.1f} us  {nb / us / 1e3:7.0f} GB/s algorithmic', flush=True)


if __name__ == '__main__':
    main()
