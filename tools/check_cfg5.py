"""cfg5 shape (BASELINE.json configs[4]): 640x384 (H x W), batch 4 per GPU, bf16: two trainer iterations run and stay finite."""
import sys, os, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd import synthetic as syn
from seg2eye_amd.options import default_opt, image_hw
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=384, aspect_ratio=0.6, batchSize=4, compute_dtype='bf16', gpu_ids=[0], hip_graphs=True)
h, w = image_hw(opt)
print('image HxW', h, w)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
b = syn.make_batch(4, h, w, seed=7)
dev = torch.device('cuda:0')
data = {'label': torch.from_numpy(b['label']).to(dev), 'style_image': torch.from_numpy(b['style_image']).to(dev),
        'target': torch.from_numpy(b['target']).to(dev), 'filename': b['filename']}
for it in range(4):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(5):
    tr.run_generator_one_step(dict(data)); tr.run_discriminator_one_step(dict(data))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print({k: float(v) for k, v in tr.get_latest_losses().items()}, 'fake', tuple(tr.get_latest_generated().shape), '%.1f ms/step, %.1f img/s' % (dt * 1e3, 4 / dt))
