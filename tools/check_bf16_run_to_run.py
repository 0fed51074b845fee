import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from conftest import load_golden, filled_state
from seg2eye_amd import networks, packing
from seg2eye_amd.options import default_opt
z = load_golden('g_ngf16_128x64')
DEV = 'cuda:0'
for dt in ('fp32', 'bf16'):
    for use_plan in (False, True):
        opt = default_opt(ngf=16, crop_size=64, aspect_ratio=0.5, compute_dtype=dt, gpu_ids=[0])
        G = networks.define_G(opt); G.load_state_dict(filled_state(z, 'G')); G.eval()
        if not use_plan:
            class Dummy:
                def __init__(self, *a): pass
                def __enter__(self): return None
                def __exit__(self, *e): return False
            orig = packing.network_scope; packing.network_scope = Dummy
        w = torch.from_numpy(z['w']).to(DEV); lab = torch.from_numpy(z['label']).to(DEV)
        ys = []
        sig = []
        with torch.no_grad():
            for it in range(4):
                ys.append(G(lab, w).float().clone())
                sig.append(G.__dict__['_sn_owned_bank'].sigma.clone())
        if not use_plan: packing.network_scope = orig
        print(dt, 'plan' if use_plan else 'noplan', ['%.2e' % float((ys[i] - ys[0]).abs().max()) for i in range(1, 4)],
              'sigma rel diff', ['%.2e' % float(((sig[i] - sig[0]) / sig[0]).abs().max()) for i in range(1, 4)])
