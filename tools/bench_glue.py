#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound glue kernels (in_stats, modulate fwd/bwd, label conv) at the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seg2eye_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (n, hw, c) in [(8, 256, 128), (8, 256, 64), (8, 128, 256), (8, 64, 512), (8, 32, 1024), (8, 16, 1024)]:
    x = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    mb = x.numel() * 2 / 1e6
    t = timeit(lambda: ops.in_stats(x))
    print('in_stats  n%d %dx%d c%d: %7.1f us  %6.2f TB/s (incl. zero+finalize launches)' % (n, hw, hw, c, t, mb / t))
for (n, hw, c) in [(8, 256, 128), (8, 256, 64), (8, 128, 256), (8, 64, 512), (8, 32, 1024)]:
    x = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16).requires_grad_(True)
    gb = torch.randn(n, hw, hw, 2 * c, device=dev).to(torch.bfloat16).requires_grad_(True)
    style = torch.randn(n, 2 * c, device=dev).requires_grad_(True)
    stats = ops.in_stats(x.detach())
    mb = x.numel() * 2 / 1e6
    t = timeit(lambda: ops.spade_style_modulate(x.detach(), gb.detach(), style.detach(), stats, True))
    print('mod_fwd   n%d %dx%d c%d: %7.1f us  %6.2f TB/s' % (n, hw, hw, c, t, 4 * mb / t))
    out = ops.spade_style_modulate(x, gb, style, stats, True)
    g = torch.randn_like(out)
    t = timeit(lambda: torch.autograd.grad(out, [x, gb, style], g, retain_graph=True))
    print('mod_bwd   n%d %dx%d c%d: %7.1f us  %6.2f TB/s (10 units; reduce+apply+style)' % (n, hw, hw, c, t, 10 * mb / t))
from seg2eye_amd import synthetic as syn
lab = torch.from_numpy(syn.ellipse_labels(8, 256, 256, seed=1)).to(dev)[:, 0].contiguous()
wsh = torch.randn(128, 4, 3, 3, device=dev) * 0.3
bsh = torch.randn(128, device=dev) * 0.1
for hw in (256, 128, 64, 16):
    t = timeit(lambda: ops.label_conv3x3_raw(lab, wsh, bsh, 8, 256, 256, hw, hw, 128, True, torch.bfloat16))
    mb = 8 * hw * hw * 128 * 2 / 1e6
    print('label_conv n8 %dx%d -> 128ch: %7.1f us  %6.2f TB/s (write only)' % (hw, hw, t, mb / t))
lab0 = torch.zeros_like(lab)
t = timeit(lambda: ops.label_conv3x3_raw(lab0, wsh, bsh, 8, 256, 256, 256, 256, 128, True, torch.bfloat16))
print('label_conv uniform labels 256: %7.1f us' % t)
labr = torch.randint(0, 4, lab.shape, device=dev, dtype=torch.uint8)
t = timeit(lambda: ops.label_conv3x3_raw(labr, wsh, bsh, 8, 256, 256, 256, 256, 128, True, torch.bfloat16))
print('label_conv random labels 256: %7.1f us' % t)
out = torch.empty(8, 256, 256, 128, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: out.fill_(1.0))
print('torch fill 134MB: %7.1f us' % t)
