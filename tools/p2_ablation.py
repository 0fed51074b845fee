#!/usr/bin/env python3
"""Timing ablation of the patch-resident conv kernels.  csrc/conv_patch2.hip reads S2E_P2_DEBUG (bits: 1 no epilogue,
2 no loads in the multiply loop, 4 no MFMAs, 8 no fragment reads, 32 no global stores, 64 no epilogue staging; bits 8.. a start
delay in microseconds per group of workgroups) -- results are WRONG with any bit set, only the time means something.

    for d in 0 1 2 4 8 14 15 32 64 96; do S2E_CONV_PATCH2=448 S2E_P2_DEBUG=$d python tools/p2_ablation.py; done
    S2E_CONV_PATCH2=0 python tools/p2_ablation.py          # the first-generation kernel on the same shapes"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops, _lib as L
dt, dev = torch.bfloat16, torch.device('cuda:0')
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n, H, cin, cout in ((8, 256, 128, 256), (8, 256, 256, 128), (8, 128, 128, 512)):
    x = torch.randn(n, H, H, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wp = ops.pack_weight(w, dt, cin, False)
    t = timeit(lambda: ops.conv2d_raw(x, wp, None, None, None, (H, H, cout), 3, 3, 1, 1))
    fl = 2.0 * n * H * H * cin * cout * 9
    print('dbg=%s patch2=%s c%d->%d @%d: %.1f us %.0f TF' % (os.environ.get('S2E_P2_DEBUG', '0'), os.environ.get('S2E_CONV_PATCH2', 'on'), cin, cout, H, t * 1e3, fl / t / 1e9), flush=True)
