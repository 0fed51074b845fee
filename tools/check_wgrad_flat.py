"""conv_wgrad_flat.hip (through s2e_conv2d_wgrad_multi) against torch fp64 on the bf16-rounded operands, and against the generic kernel's time.
  python tools/check_wgrad_flat.py            parity on small / ragged shapes, every kind
  python tools/check_wgrad_flat.py --bench    + the step's layers: one multi call per backward pass, flat on (this process) -- run again
                                              with S2E_WGRAD_FLAT=0 for the generic side"""
import sys
import os
import ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from seg2eye_amd import _lib as L

dev = 'cuda'
dt = torch.bfloat16


def desc(n, hi, wi, cin, cout, k, s, p):
    ho, wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
    return (n, hi, wi, cin, ho, wo, cout, k, k, s, p, 0, 0, 0, 0)


def make(cfg, seed):
    n, hi, wi, cin, cout, k, s, p = cfg
    d = desc(*cfg)
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(n, hi, wi, cin, generator=g).to(dev).to(dt)
    gy = torch.randn(n, d[4], d[5], cout, generator=g).to(dev).to(dt)
    return d, x, gy


def run_multi(items, bias=True, reps=0):
    """items: [(desc, x, gy)] -> [(dw (cout, k*k*cin) fp32, db)], us per call"""
    arr = (L.WgradMultiJob * len(items))()
    outs = []
    for a, (d, x, gy) in zip(arr, items):
        dw = torch.zeros(d[6], d[7] * d[8] * d[3], device=dev)
        db = torch.zeros(d[6], device=dev) if bias else None
        a.x, a.gy, a.dw, a.dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), (db.data_ptr() if bias else None)
        a.d = L.ConvDesc(*d)
        outs.append((dw, db))
    wsb = int(L.lib().s2e_conv2d_wgrad_multi_workspace_bytes(L.S2E_BF16, C.byref(arr), len(items)))
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: L.check(L.lib().s2e_conv2d_wgrad_multi(L.S2E_BF16, C.byref(arr), len(items), ws.data_ptr() if wsb else None, wsb, st), 'multi')
    call()
    torch.cuda.synchronize()
    us = None
    if reps:
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            call()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000.0 / reps
    return outs, us


def ref(d, x, gy):
    n, hi, wi, cin, ho, wo, cout, k, _, s, p = d[:11]
    xx = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, device=dev, requires_grad=True)
    y = F.conv2d(xx, w, None, s, p)
    (gw,) = torch.autograd.grad(y, w, gy.double().permute(0, 3, 1, 2))
    return gw.permute(0, 2, 3, 1).reshape(cout, k * k * cin), gy.double().sum((0, 1, 2))


def main():
    bench = '--bench' in sys.argv
    cfgs = [(2, 32, 32, 64, 128, 3, 2, 1), (3, 24, 40, 128, 72, 3, 2, 1), (2, 16, 16, 192, 256, 3, 2, 1), (5, 8, 8, 64, 64, 3, 2, 1),
            (2, 33, 33, 64, 128, 4, 2, 2), (3, 65, 37, 128, 136, 4, 2, 2), (2, 17, 17, 64, 64, 4, 2, 2),
            (2, 33, 33, 128, 256, 4, 1, 2), (3, 17, 21, 64, 72, 4, 1, 2),
            (2, 40, 24, 128, 64, 1, 1, 0), (4, 8, 8, 128, 256, 3, 1, 1), (2, 12, 20, 64, 128, 3, 1, 1)]
    kinds = [L.lib().s2e_conv2d_wgrad_multi_kind(L.S2E_BF16, C.byref(L.ConvDesc(*desc(*c)))) for c in cfgs]
    items = [make(c, 100 + i) for i, c in enumerate(cfgs)]
    outs, _ = run_multi(items)
    for c, kd, (d, x, gy), (dw, db) in zip(cfgs, kinds, items, outs):
        gw, gb = ref(d, x, gy)
        e1 = float((dw.double() - gw).abs().max() / gw.abs().max())
        e2 = float((db.double() - gb).abs().max() / gb.abs().max())
        print('n%d %dx%d c%d->%d k%d s%d  kind %d  dW rel %.2e  db rel %.2e' % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], kd, e1, e2), flush=True)
        assert (e1 < 2e-3 and e2 < 2e-3) or os.environ.get('S2E_WF_NOEPI'), (c, e1, e2)
    print('parity ok')
    if bench:
        d_step = [(16, 129, 129, 64, 128, 4, 2, 2), (16, 65, 65, 128, 256, 4, 2, 2), (16, 33, 33, 256, 512, 4, 1, 2),
                  (16, 65, 65, 64, 128, 4, 2, 2), (16, 33, 33, 128, 256, 4, 2, 2), (16, 17, 17, 256, 512, 4, 1, 2)]
        g_step = [(32, 128, 128, 64, 128, 3, 2, 1), (32, 64, 64, 128, 256, 3, 2, 1), (32, 32, 32, 256, 512, 3, 2, 1), (32, 16, 16, 512, 512, 3, 2, 1),
                  (32, 8, 8, 512, 512, 3, 2, 1)]
        ones = [(8, 256, 256, 128, 64, 1, 1, 0), (8, 128, 128, 256, 128, 1, 1, 0), (8, 64, 64, 512, 256, 1, 1, 0), (8, 32, 32, 1024, 512, 1, 1, 0)]
        small = [(8, 8, 8, 1024, 1024, 3, 1, 1), (8, 8, 8, 1024, 1024, 3, 1, 1), (8, 8, 8, 128, 2048, 3, 1, 1), (8, 8, 8, 128, 2048, 3, 1, 1)]
        for name, group in (('D step: 4x4', d_step), ('G step: netE 3x3 s2', g_step), ('1x1 shortcuts', ones), ('8x8 maps 3x3 s1', small),
                            ('G step: all', g_step + ones + small)):
            its = [make(c, 7 + i) for i, c in enumerate(group)]
            fl = sum(2.0 * d[0] * d[4] * d[5] * d[3] * d[6] * d[7] * d[8] for d, _, _ in its)
            _, us = run_multi(its, reps=20)
            print('%-22s %2d jobs  %.1f GFLOP  %.1f us  %.0f TF' % (name, len(its), fl * 1e-9, us, fl / us * 1e-6), flush=True)
            for c in group if '--each' in sys.argv else []:
                it = [make(c, 3)]
                f1 = 2.0 * it[0][0][0] * it[0][0][4] * it[0][0][5] * c[3] * c[4] * c[5] * c[5]
                _, u1 = run_multi(it, reps=20)
                print('    n%d %dx%d c%d->%d k%d s%d: %.1f us  %.0f TF' % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], u1, f1 / u1 * 1e-6), flush=True)


if __name__ == '__main__':
    main()
