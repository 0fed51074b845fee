#!/usr/bin/env python3
"""Per-shape time of the MFMA kernels inside the real G+D step (eager, HIP events per launch)."""
import sys, os, io, contextlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd import ops
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0])
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
def step():
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
for _ in range(2): step()
torch.cuda.synchronize(); prof = ops.LaunchProfiler(); ops.LaunchProfiler.install(prof)
N = 3
for _ in range(N): step()
torch.cuda.synchronize(); ops.LaunchProfiler.install(None)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for fam, fl, s, e, tag, _nb, _ex in prof.records:
    a = agg[(fam, tag)]; a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl
for fam in ('conv_patch', 'conv_plane', 'conv_igemm', 'conv_small', 'conv_wgrad_patch', 'conv_wgrad', 'conv_wgrad_small'):
    items = sorted(((k, v) for k, v in agg.items() if k[0] == fam), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in items) / N
    print('== %s total %.2f ms/step' % (fam, tot))
    for (f, tag), v in items[:40]:
        print('  %-34s x%4.1f  %6.3f ms/step  %7.1f TF' % (tag, v[0] / N, v[1] / N, v[2] / (v[1] * 1e-3) / 1e12))
print('== HBM-bound families (algorithmic bytes / HIP-event time)')
for fam, v in sorted(prof.summary().items(), key=lambda kv: -kv[1]['ms']):
    if v['flops'] == 0:
        print('  %-22s x%5.1f  %6.3f ms/step  %7.0f GB/s' % (fam, v['launches'] / N, v['ms'] / N, v['bytes'] / (v['ms'] * 1e-3) / 1e9))
