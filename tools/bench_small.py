#!/usr/bin/env python3
"""The degenerate-channel convolutions of the step (csrc/conv_small.hip): forward / data gradient / weight gradient of the
generator's image conv (64 -> 1), the PatchGAN heads (512 -> 1) and the encoder's first layer (1 -> 64), each checked against
torch's fp32 conv on the same bf16-rounded operands and timed with HIP events; GB/s = the wide tensor once."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops
from seg2eye_amd._lib import ACT_NONE, ACT_LRELU, ACT_TANH
dev = torch.device('cuda:0')
dt = torch.bfloat16


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


shapes = [  # name, n, H, W, cin, cout, k, stride, pad, in_act, out_act
    ('conv_img 64->1 @256', 8, 256, 256, 64, 1, 3, 1, 1, ACT_LRELU, ACT_TANH),
    ('D head 512->1 @34', 16, 34, 34, 512, 1, 4, 1, 2, ACT_NONE, ACT_NONE),
    ('D head 512->1 @18', 16, 18, 18, 512, 1, 4, 1, 2, ACT_NONE, ACT_NONE),
    ('E layer0 1->64 s2 @256', 32, 256, 256, 1, 64, 3, 2, 1, ACT_NONE, ACT_NONE),
]
torch.manual_seed(0)
for name, n, H, W, cin, cout, k, s, p, ia, oa in shapes:
    x = torch.randn(n, H, W, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    b = torch.randn(cout, device=dev)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = torch.randn(n, Ho, Wo, cout, device=dev).to(dt)
    wp, wpt = ops.pack_weight(w, dt, cin, False), ops.pack_weight(w, dt, cin, True)
    wide = max(x.numel(), gy.numel()) * 2 / 1e3
    # references (fp32 math on the rounded operands)
    xr = x.float().permute(0, 3, 1, 2)
    xa = F.leaky_relu(xr, 0.2) if ia == ACT_LRELU else xr
    wr = w.to(dt).float()
    yr = F.conv2d(xa, wr, b, s, p)
    if oa == ACT_TANH: yr = torch.tanh(yr)
    y = ops.conv2d_raw(x, wp, b, None, None, (Ho, Wo, cout), k, k, s, p, False, ia, oa)
    e_f = rel(y.permute(0, 3, 1, 2), yr)
    tf = t(lambda: ops.conv2d_raw(x, wp, b, None, None, (Ho, Wo, cout), k, k, s, p, False, ia, oa))
    line = '%-24s fwd %6.1f us %5.0f GB/s (rel %.1e)' % (name, tf, wide / tf, e_f)
    gr = gy.float().permute(0, 3, 1, 2)
    if s == 1:
        dxr = F.conv_transpose2d(gr, wr, None, s, p)
        dx = ops.conv2d_raw(gy, wpt, None, None, None, (H, W, cin), k, k, s, p, True)
        e_d = rel(dx.permute(0, 3, 1, 2), dxr)
        td = t(lambda: ops.conv2d_raw(gy, wpt, None, None, None, (H, W, cin), k, k, s, p, True))
        line += ' | dgrad %6.1f us %5.0f GB/s (rel %.1e)' % (td, wide / td, e_d)
    dwr = torch.nn.grad.conv2d_weight(xa, w.shape, gr, s, p)
    dw, _ = ops.conv2d_wgrad_raw(x, gy, k, k, s, p, ia)
    e_w = rel(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), dwr)
    tw = t(lambda: ops.conv2d_wgrad_raw(x, gy, k, k, s, p, ia))
    line += ' | wgrad %6.1f us %5.0f GB/s (rel %.1e)' % (tw, wide / tw, e_w)
    print(line, flush=True)
