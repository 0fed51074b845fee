import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import test_networks_gpu as T
from seg2eye_amd import ops, synthetic as syn
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
DEV = T.DEV
b1 = T._batch(2, 256, 256, 91); b2 = T._batch(2, 256, 256, 92)
res = {}
for mode in ('eager', 'eager2', 'graph', 'graph2'):
    opt = T._opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=2, compute_dtype='bf16', hip_graphs=mode.startswith('graph'))
    tr = Pix2PixTrainer(opt); m = tr.pix2pix_model
    for net, seed in ((m.netG, 1), (m.netD, 2), (m.netE, 3)):
        sd = syn.fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed)
        with torch.no_grad():
            for k, v in net.state_dict().items(): v.copy_(torch.from_numpy(sd[k]))
    imgs = []
    for it, b in enumerate((b1, b2, b1, b2)):
        tr.run_generator_one_step(dict(b)); tr.run_discriminator_one_step(dict(b))
        imgs.append(tr.get_latest_generated().detach().float().cpu().clone())
    res[mode] = imgs; del tr, m; torch.cuda.empty_cache()
for a, b in (('eager', 'eager2'), ('graph', 'graph2'), ('eager', 'graph')):
    print(a, b, [round(T._relrms(res[a][i], res[b][i]), 5) for i in range(4)])
