#!/usr/bin/env python3
"""Power-iteration chain of the style encoder's bank (6 layers, 25 MB of fp32 weights): us per iteration pair, 16 iterations
per call as in the benchmark step (one per style image, pix2pix_model.py:280-290)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import networks, spectral
from seg2eye_amd.options import default_opt
dev = torch.device('cuda:0')
opt = default_opt(gpu_ids=[0], compute_dtype='bf16')
for name, net in (('E', networks.define_E(opt)), ('D', networks.define_D(opt))):
    net = net.to(dev).train()
    bank = spectral.ensure_bank(net)
    for it in (1, 16):
        for _ in range(3): bank.step(True, it)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): bank.step(True, it)
        e.record(); torch.cuda.synchronize()
        mb = 4e-6 * sum(r * c for r, c in zip(bank.rows, bank.cols))
        print('%s bank (%d layers, %.1f MB, chain=%d), %2d iterations: %.1f us per call, %.1f us per iteration' % (
            name, bank.n, mb, bank.chain, it, s.elapsed_time(e) / 20 * 1e3, s.elapsed_time(e) / 20 / it * 1e3))
