import sys, os, io, contextlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0])
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
for it in range(2):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=12)
rows = []
for e in ka:
    dt = getattr(e, 'self_device_time_total', 0) or getattr(e, 'self_cuda_time_total', 0)
    if dt <= 0 or not e.key.startswith('aten::'):
        continue
    st = [x for x in (e.stack or []) if 'seg2eye_amd' in x]
    rows.append((dt, e.count, e.key, str(e.input_shapes)[:50], st[0].split('seg2eye_amd/')[-1][:60] if st else '?'))
for r in sorted(rows, reverse=True)[:50]:
    print('%8.1f us x%3d  %-24s %-50s %s' % r)
