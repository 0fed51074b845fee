"""conv_c8.hip (the PatchGAN's 8-channel 4x4 stride-2 first layer) through s2e_conv2d against torch fp64 on the bf16-rounded operands;
--bench: microseconds per launch at the step's two sizes (run again with S2E_CONV_C8=0 for the implicit-GEMM side)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from seg2eye_amd import ops
from seg2eye_amd.ops import conv as oc

dev, dt = 'cuda', torch.bfloat16


def t_us(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000.0 / n


def run(n, hi, wi, cin_real=5, bias=True, lrelu=True, res=False, bench=False, check=True):
    g = torch.Generator(device='cpu').manual_seed(hi * 7 + wi)
    ho, wo = hi // 2 + 1, wi // 2 + 1
    x = torch.zeros(n, hi, wi, 8)
    x[..., :cin_real] = torch.randn(n, hi, wi, cin_real, generator=g)
    x = x.to(dev).to(dt)
    w = (torch.randn(64, cin_real, 4, 4, generator=g) / (cin_real * 16) ** 0.5).to(dev)
    b = torch.randn(64, generator=g).to(dev) if bias else None
    r = torch.randn(n, ho, wo, 64, generator=g).to(dev).to(dt) if res else None
    act = ops.ACT_LRELU if lrelu else ops.ACT_NONE
    wp = oc.pack_weight(w, dt, 8, False, None)
    fn = lambda: oc.conv2d_raw(x, wp, b, r, None, (ho, wo, 64), 4, 4, 2, 2, False, ops.ACT_NONE, act)
    y = fn()
    out = 'n%d %dx%d%s%s%s' % (n, hi, wi, ' +b' if bias else '', ' +lrelu' if lrelu else '', ' +res' if res else '')
    if check:
        yr = F.conv2d(x[..., :cin_real].double().permute(0, 3, 1, 2), w.to(dt).double(), None if b is None else b.double(), 2, 2).permute(0, 2, 3, 1)
        if res:
            yr = yr + r.double()
        if lrelu:
            yr = F.leaky_relu(yr, 0.2)
        err = float((y.double() - yr).abs().max() / yr.abs().max())
        assert err < 1e-2, (out, err)
        out += '  rel %.2e' % err
    if bench:
        us = t_us(fn)
        out += '  %.1f us  %.2f TB/s' % (us, (x.numel() + y.numel()) * 2 / us * 1e-6)
    print(out, flush=True)
    # ---- data gradient
    gy = torch.randn(n, ho, wo, 64, generator=g).to(dev).to(dt)
    wpt = oc.pack_weight(w, dt, 8, True, None)
    fd = lambda: oc.conv2d_raw(gy, wpt, None, None, None, (hi, wi, 8), 4, 4, 2, 2, True, ops.ACT_NONE, ops.ACT_NONE)
    gx = fd()
    out = '   D n%d %dx%d' % (n, hi, wi)
    if check:
        op = hi + 4 - 4 - (ho - 1) * 2
        opw = wi + 4 - 4 - (wo - 1) * 2
        gr = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.to(dt).double(), None, 2, 2, output_padding=(op, opw)).permute(0, 2, 3, 1)
        err = float((gx[..., :cin_real].double() - gr).abs().max() / gr.abs().max())
        assert err < 1e-2 and float(gx[..., cin_real:].abs().max()) == 0.0, (out, err)
        out += '  rel %.2e' % err
    if bench:
        us = t_us(fd)
        out += '  %.1f us  %.2f TB/s' % (us, (gy.numel() + gx.numel()) * 2 / us * 1e-6)
    print(out, flush=True)
    # ---- weight gradient (through the multi-job call, as the trainer's flush issues it)
    import ctypes as C
    from seg2eye_amd import _lib as L
    arr = (L.WgradMultiJob * 1)()
    dw = torch.zeros(64, 128, device=dev)
    db = torch.zeros(64, device=dev)
    arr[0].x, arr[0].gy, arr[0].dw, arr[0].dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), db.data_ptr()
    arr[0].d = L.ConvDesc(n, hi, wi, 8, ho, wo, 64, 4, 4, 2, 2, 0, 0, 0, 0)
    kind = L.lib().s2e_conv2d_wgrad_multi_kind(L.S2E_BF16, C.byref(arr[0].d))
    wsb = int(L.lib().s2e_conv2d_wgrad_multi_workspace_bytes(L.S2E_BF16, C.byref(arr), 1))
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fw = lambda: L.check(L.lib().s2e_conv2d_wgrad_multi(L.S2E_BF16, C.byref(arr), 1, ws.data_ptr() if wsb else None, wsb, st), 'multi')
    fw()
    torch.cuda.synchronize()
    out = '   W n%d %dx%d kind %d' % (n, hi, wi, kind)
    if check:
        ww = torch.zeros(64, 8, 4, 4, dtype=torch.float64, device=dev, requires_grad=True)
        yy = F.conv2d(x.double().permute(0, 3, 1, 2), ww, None, 2, 2)
        (gw,) = torch.autograd.grad(yy, ww, gy.double().permute(0, 3, 1, 2))
        gw = gw.permute(0, 2, 3, 1).reshape(64, 128)
        e1 = float((dw.double() - gw).abs().max() / gw.abs().max())
        e2 = float((db.double() - gy.double().sum((0, 1, 2))).abs().max() / gy.double().sum((0, 1, 2)).abs().max())
        assert e1 < 2e-3 and e2 < 2e-3, (out, e1, e2)
        out += '  dW rel %.2e  db rel %.2e' % (e1, e2)
    if bench:
        out += '  %.1f us' % t_us(fw)
    print(out, flush=True)


if __name__ == '__main__':
    run(2, 64, 64)
    run(3, 33, 47, bias=False, lrelu=False)
    run(2, 40, 24, res=True)
    run(1, 129, 129)
    run(2, 256, 256)
    print('parity ok')
    if '--bench' in sys.argv:
        run(16, 256, 256, bench=True, check=False)
        run(16, 128, 128, bench=True, check=False)
        run(8, 256, 256, bench=True, check=False)
        run(8, 128, 128, bench=True, check=False)
        # the D step's call: both scales in one multi-job launch
        import ctypes as C
        from seg2eye_amd import _lib as L
        arr = (L.WgradMultiJob * 2)()
        keep = []
        for a, (n, h) in zip(arr, ((16, 256), (16, 128))):
            ho = h // 2 + 1
            x = torch.randn(n, h, h, 8, device=dev).to(dt)
            gy = torch.randn(n, ho, ho, 64, device=dev).to(dt)
            dw, db = torch.zeros(64, 128, device=dev), torch.zeros(64, device=dev)
            keep += [x, gy, dw, db]
            a.x, a.gy, a.dw, a.dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), db.data_ptr()
            a.d = L.ConvDesc(n, h, h, 8, ho, ho, 64, 4, 4, 2, 2, 0, 0, 0, 0)
        wsb = int(L.lib().s2e_conv2d_wgrad_multi_workspace_bytes(L.S2E_BF16, C.byref(arr), 2))
        ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        print('   W both scales, one call: %.1f us' % t_us(lambda: L.check(L.lib().s2e_conv2d_wgrad_multi(L.S2E_BF16, C.byref(arr), 2, ws.data_ptr() if wsb else None, wsb, st), 'multi')))
