import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from conftest import load_golden, filled_state
from oracle import seg2eye_oracle as O
from seg2eye_amd import networks, ops, synthetic as syn
from seg2eye_amd.options import default_opt
from seg2eye_amd.networks import architecture as A
from seg2eye_amd.networks.base_network import sn_weight
from seg2eye_amd.networks.normalization import SegMap

tag = 'g_ngf8_64'; ngf, crop, ar = 8, 64, 1.0
z = load_golden(tag); sd = filled_state(z, 'G')
H, W, sh, sw = [int(v) for v in z['hw']]
lab = torch.from_numpy(z['label'].astype(np.int64)); seg = O.one_hot_labels(lab, 4)
w = torch.from_numpy(z['w'])

ost = {}
def o_resblk(sdd, prefix, x, seg, w, training, updates):
    learned = (prefix + '.conv_s.weight_orig') in sdd
    keep = lambda n, t: (t.retain_grad(), ost.__setitem__(prefix + '.' + n, t), t)[2]
    keep('x', x)
    if learned:
        ws = O._sn_conv_weight(sdd, prefix + '.conv_s', training, updates)
        x_s = keep('xs', F.conv2d(keep('ns', O.spade_style_block(sdd, prefix + '.norm_s', x, seg, w)), ws))
    else:
        x_s = x
    w0 = O._sn_conv_weight(sdd, prefix + '.conv_0', training, updates)
    h0 = keep('h0', F.leaky_relu(O.spade_style_block(sdd, prefix + '.norm_0', x, seg, w), 0.2))
    dx = keep('dx0', F.conv2d(h0, w0, sdd[prefix + '.conv_0.bias'], padding=1))
    w1 = O._sn_conv_weight(sdd, prefix + '.conv_1', training, updates)
    h1 = keep('h1', F.leaky_relu(O.spade_style_block(sdd, prefix + '.norm_1', dx, seg, w), 0.2))
    dx1 = F.conv2d(h1, w1, sdd[prefix + '.conv_1.bias'], padding=1)
    return keep('out', x_s + dx1)
O.spade_style_resblk = o_resblk
leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
wt = w.clone().requires_grad_(True)
yo = O.generator_forward(leaf, seg, wt, sh, sw, training=False)
proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(yo.shape), seed=7))
(yo * proj).sum().backward()

gst = {}
def g_forward(self, x, seg, latent_style):
    prefix = self._dbg_name
    def keep(n, t):
        t.retain_grad(); gst[prefix + '.' + n] = t; return t
    seg = SegMap.of(seg)
    keep('x', x)
    stats = ops.in_stats(x.detach())
    if self.learned_shortcut:
        x_s = keep('xs', ops.conv2d(keep('ns', self.norm_s(x, seg, latent_style, stats, lrelu=False)), sn_weight(self.conv_s)))
    else:
        x_s = x
    h0 = keep('h0', self.norm_0(x, seg, latent_style, stats, lrelu=True))
    dx = keep('dx0', ops.conv2d(h0, sn_weight(self.conv_0), self.conv_0.bias, None, 1, 1))
    h1 = keep('h1', self.norm_1(dx, seg, latent_style, None, lrelu=True))
    return keep('out', ops.conv2d(h1, sn_weight(self.conv_1), self.conv_1.bias, x_s, 1, 1))
A.SPADE_STYLE_ResnetBlock.forward = g_forward
opt = default_opt(ngf=ngf, crop_size=crop, aspect_ratio=ar, compute_dtype='fp32', gpu_ids=[0])
G = networks.define_G(opt); G.load_state_dict(sd); G.eval()
for n, m in G.named_children():
    if isinstance(m, A.SPADE_STYLE_ResnetBlock): m._dbg_name = n
cap = {}
orig_bwd = ops.ModulateFn.backward
def bwd(ctx, g):
    x, gb, style, stats = ctx.saved_tensors
    out = orig_bwd(ctx, g)
    if tuple(x.shape) == (2, 64, 64, 8) and ctx.lrelu:
        cap.update(g=g.clone(), dx=out[0].clone(), x=x.clone())
    return out
ops.ModulateFn.backward = staticmethod(bwd)
wg = w.cuda().requires_grad_(True)
y = G(torch.from_numpy(z['label']).cuda(), wg)
(y.float() * proj.cuda()).sum().backward()
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
for blk in ('up_3', 'up_2', 'up_1', 'up_0', 'G_middle_1', 'G_middle_0', 'head_0'):
    for n in ('out', 'h1', 'dx0', 'h0', 'xs', 'ns', 'x'):
        k = blk + '.' + n
        if k not in ost: continue
        a, b = gst[k], ost[k]
        fa = a.detach().permute(0, 3, 1, 2).cpu(); ga = a.grad.permute(0, 3, 1, 2).cpu()
        print('%-16s fwd %.2e  grad %.2e   |grad|max %.2e' % (k, rel(fa, b.detach()), rel(ga, b.grad), float(b.grad.abs().max())))

nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
print('captured g  vs my h1.grad     ', rel(nchw(cap['g']), nchw(gst['up_3.h1'].grad)))
print('captured g  vs oracle h1.grad ', rel(nchw(cap['g']), ost['up_3.h1'].grad))
print('captured dx vs my dx0.grad    ', rel(nchw(cap['dx']), nchw(gst['up_3.dx0'].grad)))
print('captured dx vs oracle dx0.grad', rel(nchw(cap['dx']), ost['up_3.dx0'].grad))
print('captured x  vs oracle dx0     ', rel(nchw(cap['x']), ost['up_3.dx0'].detach()))
d = (nchw(cap['dx']) - ost['up_3.dx0'].grad).abs()
print('where: per-sample max', d.amax(dim=(1,2,3)), 'per-channel max', d.amax(dim=(0,2,3)))
idx = torch.nonzero(d > 0.3 * d.max())
print('n bad', idx.shape[0], idx[:10].tolist())
