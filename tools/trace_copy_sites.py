#!/usr/bin/env python3
"""Which Python call sites issue the device-to-device copies of one eager G+D step (Tensor.copy_ / clone / contiguous / to / cat)."""
import collections, contextlib, io, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=False)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
def step():
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
for _ in range(2):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
def wrap(name, fn):
    def f(*a, **k):
        st = [s for s in traceback.extract_stack()[:-1] if 'seg2eye_amd' in s.filename]
        t = a[0] if isinstance(a[0], torch.Tensor) else (a[0][0] if a[0] else None)
        if st and t is not None and t.is_cuda:
            s = st[-1]
            sites[(name, os.path.basename(s.filename), s.lineno, s.line[:80], tuple(t.shape), str(t.dtype))] += 1
        return fn(*a, **k)
    return f
for nm in ('copy_', 'clone', 'contiguous', 'to', 'float', 'zero_', 'fill_'):
    setattr(torch.Tensor, nm, wrap(nm, getattr(torch.Tensor, nm)))
torch.cat = wrap('cat', torch.cat)
step()
torch.cuda.synchronize()
for k, c in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][2])):
    print(c, k)
