import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
graphs = '--graphs' in sys.argv
ngf = 64
opt = default_opt(ngf=ngf, ndf=ngf, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=graphs)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
for it in range(4):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
    torch.cuda.synchronize()
    m = tr.pix2pix_model
    sig = {n: [round(float(x), 4) for x in net.__dict__['_sn_owned_bank'].sigma[:4]] for n, net in (('G', m.netG), ('D', m.netD), ('E', m.netE))}
    print(it, {k: round(float(v.float().mean()), 4) for k, v in tr.get_latest_losses().items()}, sig, flush=True)
