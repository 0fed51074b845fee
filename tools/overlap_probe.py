#!/usr/bin/env python3
"""Does running a layer's weight gradient on a second HIP stream, next to its data gradient, buy anything?  For a few of the
step's (data-gradient, weight-gradient) pairs: K pairs back to back on one stream against the same K pairs with the weight
gradients forked to a side stream (event fork before each pair, one join at the end), both eager and replayed as a hipGraph
captured from the two streams.

    python tools/overlap_probe.py [--pairs 8] [--reps 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops                                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    # (n, h, w, cin, cout, k, stride, pad)
    shapes = [(8, 256, 256, 128, 256, 3, 1, 1), (8, 128, 128, 128, 512, 3, 1, 1), (8, 64, 64, 512, 256, 3, 1, 1), (8, 16, 16, 1024, 1024, 3, 1, 1),
              (8, 32, 32, 512, 512, 3, 1, 1), (16, 65, 65, 128, 256, 4, 2, 2), (32, 64, 64, 128, 256, 3, 2, 1), (8, 8, 8, 1024, 1024, 3, 1, 1)]
    side = torch.cuda.Stream()
    for (n, h, w, cin, cout, k, s, pad) in shapes:
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        x = torch.randn(n, h, w, cin, device=dev).bfloat16()
        gy = torch.randn(n, ho, wo, cout, device=dev).bfloat16()
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.02
        wpt = ops.pack_weight(wt, torch.bfloat16, cin, True)
        dws = [torch.zeros(cout, k * k * cin, device=dev) for _ in range(args.pairs)]
        gxs = [torch.empty(n, h, w, cin, device=dev, dtype=torch.bfloat16) for _ in range(args.pairs)]

        def dgrad(i):
            ops.conv2d_raw(gy, wpt, None, None, None, (h, w, cin), k, k, s, pad, True, out=gxs[i])

        def wgrad(i):
            ops.conv2d_wgrad_raw(x, gy, k, k, s, pad, dw_out=dws[i])

        def serial():
            for i in range(args.pairs):
                dgrad(i)
                wgrad(i)

        def forked():
            main_s = torch.cuda.current_stream()
            for i in range(args.pairs):
                if i == 0:
                    side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    wgrad(i)
                dgrad(i)
            main_s.wait_stream(side)

        def only(fn):
            def run():
                for i in range(args.pairs):
                    fn(i)
            return run

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.reps / args.pairs

        def graphed(fn):
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    fn()
            torch.cuda.synchronize()
            return timed(g.replay)

        res = dict(d=graphed(only(dgrad)), w=graphed(only(wgrad)), serial=graphed(serial))
        try:
            res['forked'] = graphed(forked)
        except Exception as e:                               # (a capture the runtime refuses is an answer too)
            res['forked'] = float('nan')
            print('   forked capture failed:', repr(e)[:200], flush=True)
            torch.cuda.synchronize()
        res['eager_forked'] = timed(forked)
        print(f'n{n} {h}x{w} c{cin}->{cout} k{k} s{s}: dgrad {res["d"]:7.1f}  wgrad {res["w"]:7.1f}  pair serial {res["serial"]:7.1f}  '
              f'pair forked (graph) {res["forked"]:7.1f}  forked (eager) {res["eager_forked"]:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
