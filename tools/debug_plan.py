import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0])
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
m = tr.pix2pix_model
for it in range(3):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
    for name, net in (('G', m.netG), ('D', m.netD), ('E', m.netE)):
        p = net.__dict__.get('_pack_plan')
        print(it, name, None if p is None else (len(p.jobs), p.hits, p.generation, p.dirty, sum(1 for j in p.jobs.values() if j['transposed'])))
