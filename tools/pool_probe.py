import sys, os, io, contextlib
sys.path.insert(0, '/root/repo')
import torch, bench
from seg2eye_amd import ops
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0])
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
takes = []
orig = ops.ZeroPool.take.__func__
def take(cls, numel, dtype, device):
    import traceback
    st = traceback.extract_stack(limit=4)
    takes.append((numel * torch.empty((), dtype=dtype).element_size(), ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in st[:-1][-3:])))
    return orig(cls, numel, dtype, device)
for _ in range(2):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
ops.ZeroPool.take = classmethod(take)
tr.run_generator_one_step(dict(data)); n_g = len(takes)
tr.run_discriminator_one_step(dict(data))
torch.cuda.synchronize()
print('pool high', tr.pool.high, 'cap', tr.pool.cap)
import collections
for part, name in ((takes[:n_g], 'G'), (takes[n_g:], 'D')):
    agg = collections.defaultdict(lambda: [0, 0])
    for b, where in part:
        agg[where][0] += 1; agg[where][1] += b
    print(name, 'takes', len(part), 'bytes', sum(b for b, _ in part))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print('   %8.1f MB x%3d  %s' % (v[1] / 1e6, v[0], k))
