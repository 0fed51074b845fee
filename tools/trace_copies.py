#!/usr/bin/env python3
"""Device-side copies and fills of one replayed (hipGraph) G+D step, by size: which of them are left and how large they are."""
import collections, contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
graphs = '--eager' not in sys.argv
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=graphs)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
def step():
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
for _ in range(4):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        nm = e.name
        if 'Memcpy' in nm or 'Memset' in nm or 'copyBuffer' in nm or 'at::native' in nm or 'Cijk' in nm or 'fillBuffer' in nm:
            a = agg[nm[:110]]
            a[0] += 1; a[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%6.1f/step %8.1f us/step  %s' % (c / N, t / N, k))
