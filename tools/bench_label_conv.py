"""s2e_label_conv3x3 alone, per map size (N = 8, 128 output channels, 4 classes, synthetic ellipse labels): microseconds and GB/s of stores."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from seg2eye_amd.ops import spade as sp

dev = torch.device('cuda:0')
data = bench.make_data(8, 256, 1234, dev)
label = data['label'].reshape(8, 256, 256).to(torch.uint8).contiguous()
w = torch.randn(128, 4, 3, 3, device=dev)
b = torch.randn(128, device=dev)
for h in (256, 128, 64, 32, 16):
    fn = lambda: sp.label_conv3x3_raw(label, w, b, 8, 256, 256, h, h, 128, True, torch.bfloat16)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 20
    print('%3dx%-3d  %6.1f us  %6.0f GB/s of stores' % (h, h, us, 8 * h * h * 256 / us * 1e-3))
