import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd import ops, spectral
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
REC = []; ON = [False]
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        r = f(*a, **k)
        if ON[0] and torch.is_tensor(r): REC.append((name, tuple(r.shape), r))
        return r
    setattr(mod, name, g)
for n in ('pack_weight', 'conv2d_raw', 'in_stats', 'label_conv3x3_raw', 'conv2d_wgrad_raw', 'colsum', 'onehot_nhwc_raw'):
    wrap(ops, n)
ostep = spectral.SpectralBank.step
def step(self, training, iterations=1):
    ostep(self, training, iterations)
    if ON[0]: REC.append(('sigma[%d layers]' % self.n, (self.n,), self.sigma)); REC.append(('uv_snap', tuple(self.uv_snap.shape), self.uv_snap))
spectral.SpectralBank.step = step
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
tr.pix2pix_model.train()
# manual capture of graph_G only, recording intermediates
tr._static = {k: data[k].clone() for k in ('label', 'style_image', 'target')}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): tr._g_body(tr._static); tr._d_body(tr._static)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(); ON[0] = True
with torch.cuda.graph(g): tr._g_body(tr._static)
ON[0] = False
print('recorded', len(REC))
def first_bad():
    for i, (n, s, t) in enumerate(REC):
        if not bool(torch.isfinite(t.float()).all()): return i, n, s
    return None
bE = tr.pix2pix_model.netE.__dict__['_sn_owned_bank']
def snap():
    torch.cuda.synchronize()
    return {k: getattr(bE, k).detach().cpu().clone() for k in ('table_dev', 'block_map', 'uv_arena', 'scratch')}
def diff(a, b, tag):
    out = []
    for k in a:
        same = bool((a[k] == b[k]).all()) if a[k].dtype != torch.float32 else bool(torch.equal(torch.nan_to_num(a[k]), torch.nan_to_num(b[k])))
        out.append('%s:%s nan=%d zeros=%d/%d' % (k, 'same' if same else 'CHANGED', int(torch.isnan(b[k].float()).sum()), int((b[k] == 0).sum()), b[k].numel()))
    print(tag, ' | '.join(out), flush=True)
s0 = snap(); diff(s0, s0, 'before      ')
g.replay(); s1 = snap(); diff(s0, s1, 'after replay0'); print('   first non-finite:', first_bad())
g.replay(); s2 = snap(); diff(s1, s2, 'after replay1'); print('   first non-finite:', first_bad())
w = tr.pix2pix_model.netE.layer0[0].weight_orig
print('E layer0 W finite', bool(torch.isfinite(w).all()), 'absmax', float(w.abs().max()))
