import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden, filled_state
from seg2eye_amd import networks, ops, synthetic as syn
from seg2eye_amd.options import default_opt
z = load_golden('g_ngf8_64'); sd = filled_state(z, 'G')
cap = {}
orig_bwd = ops.ModulateFn.backward
def bwd(ctx, g):
    x, gb, style, stats = ctx.saved_tensors
    out = orig_bwd(ctx, g)
    if tuple(x.shape) == (2, 64, 64, 8) and ctx.lrelu:
        cap.update(x=x.clone(), gb=gb.clone(), style=style.clone(), stats=stats.clone(), g=g.clone(), dx=out[0].clone(), dgb=out[1].clone(), dstyle=out[2].clone())
    return out
ops.ModulateFn.backward = staticmethod(bwd)
opt = default_opt(ngf=8, crop_size=64, compute_dtype='fp32', gpu_ids=[0])
G = networks.define_G(opt); G.load_state_dict(sd); G.eval()
wg = torch.from_numpy(z['w']).cuda().requires_grad_(True)
y = G(torch.from_numpy(z['label']).cuda(), wg)
proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(y.shape), seed=7)).cuda()
(y.float() * proj).sum().backward()
x, gb, style, g = [cap[k].double().cpu() for k in ('x', 'gb', 'style', 'g')]
print('g contiguous?', cap['g'].is_contiguous(), cap['g'].stride(), 'x stride', cap['x'].stride())
xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True); gbr = gb.permute(0, 3, 1, 2).clone().requires_grad_(True); sr = style.clone().requires_grad_(True)
C = 8
yr = F.leaky_relu(0.5 * (F.instance_norm(xr, eps=1e-5) * (1 + gbr[:, :C]) + gbr[:, C:] + xr * (1 + sr[:, :C, None, None]) + sr[:, C:, None, None]), 0.2)
yr.backward(g.permute(0, 3, 1, 2))
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
print('captured dx vs fp64 ref   ', rel(cap['dx'].double().cpu().permute(0, 3, 1, 2), xr.grad))
print('captured dgb vs fp64 ref  ', rel(cap['dgb'].double().cpu().permute(0, 3, 1, 2), gbr.grad))
print('captured dstyle vs fp64   ', rel(cap['dstyle'].double().cpu(), sr.grad))
st = cap['stats'].double().cpu()
print('stats mean err', rel(st[..., 0], x.mean(dim=(1, 2))), 'rstd err', rel(st[..., 1], 1 / torch.sqrt(x.var(dim=(1, 2), unbiased=False) + 1e-5)), 'rstd max', float(st[..., 1].max()))
# replay standalone
xg = cap['x'].clone().requires_grad_(True); gbg = cap['gb'].clone().requires_grad_(True); sg = cap['style'].clone().requires_grad_(True)
ops.ModulateFn.backward = staticmethod(orig_bwd)
y2 = ops.ModulateFn.apply(xg, gbg, sg, ops.in_stats(xg.detach()), True)
y2.backward(cap['g'].contiguous())
print('replayed dx vs fp64 ref   ', rel(xg.grad.double().cpu().permute(0, 3, 1, 2), xr.grad))
print('replay with strided g     ')
