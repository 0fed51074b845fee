import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from conftest import load_golden, filled_state
from oracle import seg2eye_oracle as O
from seg2eye_amd import networks, synthetic as syn
from seg2eye_amd.options import default_opt
tag = sys.argv[1] if len(sys.argv) > 1 else 'g_ngf8_64'
ngf, crop, ar = (8, 64, 1.0) if tag == 'g_ngf8_64' else (16, 64, 0.5)
z = load_golden(tag)
sd = filled_state(z, 'G')
H, W, sh, sw = [int(v) for v in z['hw']]
lab = torch.from_numpy(z['label'].astype(np.int64))
seg = O.one_hot_labels(lab, 4)
w = torch.from_numpy(z['w'])
leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
wt = w.clone().requires_grad_(True)
yo = O.generator_forward(leaf, seg, wt, sh, sw, training=False)
proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(yo.shape), seed=7))
(yo * proj).sum().backward()
opt = default_opt(ngf=ngf, crop_size=crop, aspect_ratio=ar, compute_dtype='fp32', gpu_ids=[0])
G = networks.define_G(opt); G.load_state_dict(sd); G.eval()
wg = w.cuda().requires_grad_(True)
y = G(torch.from_numpy(z['label']).cuda(), wg)
print('fwd err', float((y.float().cpu() - yo.detach()).abs().max()))
(y.float() * proj.cuda()).sum().backward()
print('grad_w per-sample rel err', [(float((wg.grad[i].cpu() - wt.grad[i]).abs().max() / wt.grad[i].abs().max())) for i in range(w.shape[0])])
rows = []
for k, p in G.named_parameters():
    a, b = p.grad.cpu(), leaf[k].grad
    rows.append((float((a - b).abs().max() / (b.abs().max() + 1e-12)), k, float(b.abs().max())))
order = [k for k, _ in G.named_parameters()]
for r in rows:
    print('%.3e  %-50s  max|g|=%.3e' % r)
