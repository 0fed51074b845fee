import sys, os, io, contextlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from seg2eye_amd import _lib
from seg2eye_amd.options import default_opt
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=False)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
bench.fill_weights(tr.pix2pix_model)
data = bench.make_data(8, 256, 1234, torch.device('cuda:0'))
tr.run_generator_one_step(dict(data)); torch.cuda.synchronize()
def rng(t): return (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
R = {k: rng(getattr(tr.optimizer_G, k)) for k in ('flat_p', 'flat_g', 'flat_m', 'flat_v')}
def where(p):
    for k, (a, b) in R.items():
        if a <= p < b: return k
    return '?'
m = tr.pix2pix_model
for n, net in (('E', m.netE), ('G', m.netG)):
    b = net.__dict__['_sn_owned_bank']
    raw = b.table_dev.cpu().numpy().tobytes()
    tab = (_lib.SnLayer * b.n).from_buffer_copy(raw)
    ua, sa = rng(b.uv_arena), rng(b.scratch)
    for i in range(min(b.n, 6)):
        L = tab[i]
        c = b.convs[i]
        print(n, i, 'rows', L.rows, 'cols', L.cols, 'w in', where(L.w), 'w==param', L.w == c.weight_orig.data_ptr(),
              'u ok', ua[0] <= L.u < ua[1], 'v ok', ua[0] <= L.v < ua[1], 't ok', sa[0] <= L.t < sa[1], 's ok', sa[0] <= L.s < sa[1],
              'param in', where(c.weight_orig.data_ptr()))
