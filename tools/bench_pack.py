#!/usr/bin/env python3
"""Weight pack from torch-order (OIHW) and channels-last fp32 masters: forward and transposed, bf16 out."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for shape in [(1024, 1024, 3, 3), (2048, 128, 3, 3), (256, 128, 3, 3), (512, 256, 4, 4)]:
    w = torch.randn(*shape, device=dev)
    wcl = w.contiguous(memory_format=torch.channels_last)
    assert not wcl.is_contiguous()
    for tr in (False, True):
        a = ops.pack_weight(w, torch.bfloat16, None, tr)
        b = ops.pack_weight(wcl, torch.bfloat16, None, tr)
        assert torch.equal(a, b), (shape, tr)
        ta, tb = t(lambda: ops.pack_weight(w, torch.bfloat16, None, tr)), t(lambda: ops.pack_weight(wcl, torch.bfloat16, None, tr))
        nb = w.numel() * 6 / 1e3
        print('%s %s: OIHW %.1f us (%.0f GB/s)   channels-last %.1f us (%.0f GB/s)' % (shape, 'tr ' if tr else 'fwd', ta, nb / ta, tb, nb / tb))
