import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden, filled_state
from oracle import seg2eye_oracle as O
from seg2eye_amd import synthetic as syn
from seg2eye_amd.options import default_opt
from seg2eye_amd.networks.architecture import SPADE_STYLE_ResnetBlock
from seg2eye_amd.networks.normalization import SegMap
z = load_golden('modules')
opt = default_opt(ngf=8, crop_size=64, compute_dtype='fp32', gpu_ids=[0])
for name, fin, fout in (('res_diff', 16, 8), ('res_same', 16, 16)):
    sd = filled_state(z, name)
    for hw in (16, 32, 64):
        lab = torch.from_numpy(syn.ellipse_labels(2, 64, 64, seed=11))
        seg = O.one_hot_labels(lab.long(), 4)
        x = torch.from_numpy(syn.hash_normal('dbg_x', (2, fin, hw, hw), seed=hw))
        w = torch.from_numpy(syn.hash_normal('dbg_w', (2, 16), seed=3))
        leaf = {('p.' + k): (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True)
        yo = O.spade_style_resblk(leaf, 'p', xo, seg, wo, False, None)
        proj = torch.from_numpy(syn.hash_uniform('dbg_p', tuple(yo.shape), seed=5))
        (yo * proj).sum().backward()
        m = SPADE_STYLE_ResnetBlock(fin, fout, opt).cuda(); m.load_state_dict(sd); m.eval()
        xg = x.cuda().permute(0, 2, 3, 1).contiguous().requires_grad_(True); wg = w.cuda().requires_grad_(True)
        y = m(xg, SegMap.of(lab.cuda()), wg)
        (y * proj.cuda().permute(0, 2, 3, 1)).sum().backward()
        rel = lambda a, b: float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
        print(name, hw, 'y %.2e dx %.2e dw %.2e' % (rel(y.permute(0, 3, 1, 2).detach(), yo.detach()), rel(xg.grad.permute(0, 3, 1, 2), xo.grad), rel(wg.grad, wo.grad)))
        worst = sorted(((rel(p.grad, leaf['p.' + k].grad), k) for k, p in m.named_parameters()), reverse=True)[:4]
        print('    worst params:', ['%s %.2e' % (k, e) for e, k in worst])
