import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
ngf = int(os.environ.get('NGF', '64'))
opt = default_opt(ngf=ngf, ndf=ngf, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
m = tr.pix2pix_model
def nn(t): return int(torch.isnan(t).sum())
def report(tag):
    torch.cuda.synchronize()
    b = {n: net.__dict__['_sn_owned_bank'] for n, net in (('G', m.netG), ('D', m.netD), ('E', m.netE))}
    print(tag, 'pG', nn(tr.optimizer_G.flat_p), 'gG', nn(tr.optimizer_G.flat_g), 'mG', nn(tr.optimizer_G.flat_m), 'vG', nn(tr.optimizer_G.flat_v),
          'pD', nn(tr.optimizer_D.flat_p), 'gD', nn(tr.optimizer_D.flat_g),
          'uv', {k: nn(v.uv_arena) for k, v in b.items()}, 'sig', {k: nn(v.sigma) for k, v in b.items()},
          'hyperG', [round(float(x), 6) for x in tr.optimizer_G.hyper], flush=True)
tr.pix2pix_model.train()
for it in range(2):
    tr._stage_inputs(dict(data)); tr.graph_G.replay(); report('it%d after graph_G' % it)
    tr.optimizer_G.step(grad_scale=1.0); report('it%d after adam_G ' % it)
    tr.graph_D.replay(); report('it%d after graph_D' % it)
    tr.optimizer_D.step(grad_scale=1.0); report('it%d after adam_D ' % it)
g = tr.optimizer_G.flat_g
