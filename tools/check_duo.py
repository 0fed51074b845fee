#!/usr/bin/env python3
"""The duo patch kernel (csrc/conv_duo.hip: two staggered workgroups per CU; S2E_CONV_DUO = minimum work items) against torch's own convolution at the bench shapes: forward with bias +
residual + LeakyReLU, data-gradient with the ReLU mask, and the fused [gamma | beta] conv + SPADE+Style modulation (dense and
through a rectangle list with an ODD number of rectangles), each timed with HIP events.
    S2E_CONV_DUO=512 python tools/check_duo.py      # S2E_CONV_DUO=0 runs the same checks on conv_patch.hip"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops  # noqa: E402
from seg2eye_amd import _lib as L  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rel(a, ref):
    return float((a.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))


def main():
    dt, dev = torch.bfloat16, torch.device('cuda:0')
    torch.manual_seed(0)
    worst = 0.0
    # ---- plain conv: (N, H, Cin, Cout)
    for n, H, cin, cout in ((8, 256, 128, 256), (8, 128, 256, 128), (8, 64, 512, 256), (8, 256, 64, 128), (8, 256, 128, 64), (8, 256, 64, 64), (8, 128, 128, 512), (3, 96, 96, 160)):
        x = torch.randn(n, H, H, cin, device=dev).to(dt)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        b = torch.randn(cout, device=dev)
        res = torch.randn(n, H, H, cout, device=dev).to(dt)
        wp, wpt = ops.pack_weight(w, dt, cin, False), ops.pack_weight(w, dt, cin, True)
        wq = w.to(dt).float()
        d = L.ConvDesc(n, H, H, cin, H, H, cout, 3, 3, 1, 1, 0, 0, 1, 0)
        kind = L.lib().s2e_conv2d_kernel_kind(L.S2E_BF16, C.byref(d))
        y = ops.conv2d_raw(x, wp, b, res, None, (H, H, cout), 3, 3, 1, 1, False, L.ACT_NONE, L.ACT_LRELU)
        ref = F.leaky_relu(F.conv2d(x.float().permute(0, 3, 1, 2), wq, b, padding=1).permute(0, 2, 3, 1) + res.float(), 0.2)
        e1 = rel(y, ref)
        gy = torch.randn(n, H, H, cout, device=dev).to(dt)
        gx = ops.conv2d_raw(gy, wpt, None, None, x, (H, H, cin), 3, 3, 1, 1, True, L.ACT_NONE, L.ACT_NONE, L.AUX_RELU_MASK)
        refg = F.conv_transpose2d(gy.float().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1) * (x.float() > 0)
        e2 = rel(gx, refg)
        fl = 2.0 * n * H * H * cin * cout * 9
        tf = timeit(lambda: ops.conv2d_raw(x, wp, b, None, None, (H, H, cout), 3, 3, 1, 1))
        td = timeit(lambda: ops.conv2d_raw(gy, wpt, None, None, None, (H, H, cin), 3, 3, 1, 1, True))
        print('conv n%d %dx%d c%d->%d kind %d | F %7.1f us %6.1f TF err %.1e | D %7.1f us %6.1f TF err %.1e'
              % (n, H, H, cin, cout, kind, tf * 1e3, fl / tf / 1e9, e1, td * 1e3, fl / td / 1e9, e2), flush=True)
        worst = max(worst, e1, e2)
    # ---- fused [gamma | beta] conv + modulation
    for n, H, c, up in ((8, 256, 128, 0), (8, 128, 256, 8), (8, 64, 512, 0), (8, 256, 64, 8)):
        nh = 128
        actv = torch.relu(torch.randn(n, H, H, nh, device=dev)).to(dt)
        w = torch.randn(2 * c, nh, 3, 3, device=dev) / (nh * 9) ** 0.5
        b = torch.randn(2 * c, device=dev) * 0.1
        hx = H // 2 if up else H
        x = torch.randn(n, hx, hx, c, device=dev).to(dt)
        stats = ops.in_stats(x)
        style = torch.randn(n, 2 * c, device=dev) * 0.3
        wp = ops.pack_weight(w, dt, nh, False)
        out = torch.empty(n, H, H, c, dtype=dt, device=dev)
        gam = torch.empty_like(out)
        st = torch.cuda.current_stream().cuda_stream

        def run(rect_list=None, count=None, o=out, g=gam):
            if rect_list is None:
                L.check(L.lib().s2e_spade_conv_modulate(L.S2E_BF16, actv.data_ptr(), wp.data_ptr(), b.data_ptr(), x.data_ptr(), stats.data_ptr(),
                                                        style.data_ptr(), 0, o.data_ptr(), g.data_ptr(), n, H, H, c, nh, 1, up, st), 'fused')
            else:
                L.check(L.lib().s2e_spade_conv_modulate_sparse(L.S2E_BF16, actv.data_ptr(), wp.data_ptr(), b.data_ptr(), x.data_ptr(), stats.data_ptr(),
                                                               style.data_ptr(), 0, o.data_ptr(), g.data_ptr(), n, H, H, c, nh, 1, up,
                                                               rect_list.data_ptr(), count.data_ptr(), st), 'fused sparse')
        run()
        gb = F.conv2d(actv.float().permute(0, 3, 1, 2), w.to(dt).float(), b, padding=1).permute(0, 2, 3, 1)
        ga, be = gb[..., :c], gb[..., c:]
        xf = x.float()
        if up:
            xf = xf.repeat_interleave(2, 1).repeat_interleave(2, 2)
        mu, rs = stats[:, :, 0].view(n, 1, 1, c), stats[:, :, 1].view(n, 1, 1, c)
        ref = F.leaky_relu(0.5 * ((xf - mu) * rs * (1 + ga) + be + xf * (1 + style[:, :c].view(n, 1, 1, c)) + style[:, c:].view(n, 1, 1, c)), 0.2)
        e1, e2 = rel(out, ref), rel(gam, ga)
        # through a rectangle list: every other rectangle, an odd number of them; the others must stay untouched
        tw, th = C.c_int(0), C.c_int(0)
        L.lib().s2e_spade_conv_modulate_rect(L.S2E_BF16, n, H, H, c, nh, up, C.byref(tw), C.byref(th))
        rects = n * ((H + th.value - 1) // th.value) * ((H + tw.value - 1) // tw.value)
        ids = torch.arange(0, rects, 2, dtype=torch.int32, device=dev)[:-1] if (rects // 2) % 2 == 0 else torch.arange(0, rects, 2, dtype=torch.int32, device=dev)
        cnt = torch.tensor([ids.numel(), 0], dtype=torch.int32, device=dev)
        out2 = torch.full_like(out, 7.0)
        gam2 = torch.full_like(out, 7.0)
        run(ids, cnt, out2, gam2)
        ty, tx = (H + th.value - 1) // th.value, (H + tw.value - 1) // tw.value
        mask = torch.zeros(rects, dtype=torch.bool, device=dev)
        mask[ids.long()] = True
        m = mask.view(n, ty, tx).repeat_interleave(th.value, 1).repeat_interleave(tw.value, 2)[:, :H, :H].unsqueeze(-1)
        e3 = float(((out2.float() - torch.where(m, ref, torch.full_like(ref, 7.0))).abs().max()) / ref.abs().max())
        fl = 2.0 * n * H * H * nh * 2 * c * 9
        t = timeit(run)
        print('fused n%d %dx%d C=%d up=%d rect %dx%d (%d listed of %d) | %7.1f us %6.1f TF | err out %.1e gamma %.1e sparse %.1e'
              % (n, H, H, c, up, tw.value, th.value, ids.numel(), rects, t * 1e3, fl / t / 1e9, e1, e2, e3), flush=True)
        worst = max(worst, e1, e2, e3)
    print('worst relative error %.2e' % worst)
    if worst > 2e-2:
        sys.exit('FAILED')


if __name__ == '__main__':
    main()
