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
def nn(t): return int(torch.isnan(t).sum())
tr.pix2pix_model.train()
tr._stage_inputs(dict(data))
mode = sys.argv[1]
if mode == 'gg':            # G graph twice, nothing in between
    for i in range(3):
        tr.graph_G.replay(); torch.cuda.synchronize(); print('G replay', i, 'nan grads', nn(tr.optimizer_G.flat_g), {k: round(float(v.float().mean()), 4) for k, v in tr.g_losses.items()})
elif mode == 'gdg':         # dirty the pool with the D graph, no optimizer steps at all
    for i in range(3):
        tr.graph_G.replay(); torch.cuda.synchronize(); print('G replay', i, 'nan grads', nn(tr.optimizer_G.flat_g))
        tr.graph_D.replay(); torch.cuda.synchronize(); print('D replay', i, 'nan grads', nn(tr.optimizer_D.flat_g))
elif mode == 'poison':      # G graph, then poison all free pool memory with NaN via a big eager alloc? (not in pool) -> skip
    pass
if mode in ('adamG', 'adamD', 'both'):
    for i in range(3):
        tr.graph_G.replay(); torch.cuda.synchronize()
        print('G replay', i, 'nan grads', nn(tr.optimizer_G.flat_g), {k: round(float(v.float().mean()), 4) for k, v in tr.g_losses.items()},
              'fake nan', nn(tr.generated), 'sigG', [round(float(x), 3) for x in tr.pix2pix_model.netG.__dict__['_sn_owned_bank'].sigma[:3]])
        if mode in ('adamG', 'both'):
            tr.optimizer_G.step(); torch.cuda.synchronize(); print('   adamG -> nan p', nn(tr.optimizer_G.flat_p), 'absmax p', float(tr.optimizer_G.flat_p.abs().max()), 'absmax g', float(tr.optimizer_G.flat_g.abs().max()))
        tr.graph_D.replay(); torch.cuda.synchronize(); print('D replay', i, 'nan grads', nn(tr.optimizer_D.flat_g), {k: round(float(v.float().mean()), 4) for k, v in tr.d_losses.items()})
        if mode in ('adamD', 'both'):
            tr.optimizer_D.step(); torch.cuda.synchronize(); print('   adamD -> nan p', nn(tr.optimizer_D.flat_p), 'absmax p', float(tr.optimizer_D.flat_p.abs().max()), 'absmax g', float(tr.optimizer_D.flat_g.abs().max()))
if mode in ('perturb', 'lr0', 'fillg'):
    for i in range(3):
        tr.graph_G.replay(); torch.cuda.synchronize()
        print('G replay', i, 'nan grads', nn(tr.optimizer_G.flat_g), 'fake nan', nn(tr.generated), {k: round(float(v.float().mean()), 4) for k, v in tr.g_losses.items()})
        if mode == 'perturb':
            tr.optimizer_G.flat_p.add_(1e-4)
        elif mode == 'lr0':
            tr.optimizer_G.param_groups[0]['lr'] = 0.0
            tr.optimizer_G.step()
        elif mode == 'fillg':
            tr.optimizer_G.flat_m.add_(1.0)          # touch a big unrelated persistent buffer eagerly
        torch.cuda.synchronize()
if mode == 'addr':
    def rng(t): return (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
    named = {}
    for tag, o in (('G', tr.optimizer_G), ('D', tr.optimizer_D)):
        for k in ('flat_p', 'flat_g', 'flat_m', 'flat_v', 'hyper'):
            named['opt%s.%s' % (tag, k)] = rng(getattr(o, k))
    m = tr.pix2pix_model
    for n, net in (('G', m.netG), ('D', m.netD), ('E', m.netE)):
        b = net.__dict__['_sn_owned_bank']
        for k in ('uv_arena', 'scratch', 'table_dev', 'block_map', 'sigma', 'uv_snap'):
            named['bank%s.%s' % (n, k)] = rng(getattr(b, k))
    named['generated'] = rng(tr.generated)
    for k, v in tr.g_losses.items(): named['loss.' + k] = rng(v)
    for k, v in tr._static.items(): named['static.' + k] = rng(v)
    items = sorted(named.items(), key=lambda kv: kv[1][0])
    for k, (a, b) in items:
        print('%-22s %#x - %#x  (%d B)' % (k, a, b, b - a))
    for i, (k, (a, b)) in enumerate(items):
        for k2, (a2, b2) in items[i + 1:]:
            if a2 < b and k.split('.')[0] != k2.split('.')[0]:
                print('OVERLAP', k, k2)
