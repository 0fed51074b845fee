"""conv_plane.hip against torch (fp64 of the bf16-rounded operands) and against the generic kernels' time, per mode.
  python tools/check_plane.py            parity on small shapes
  python tools/check_plane.py --bench    + the step's shapes, microseconds per launch: plane vs s2e_conv2d (generic / patch)"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from seg2eye_amd import ops
from seg2eye_amd.ops import conv as oc

dev = 'cuda'
dt = torch.bfloat16


def ref_fwd(x, w, b, stride, pad, res=None, lrelu=False):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None if b is None else b.double(), stride, pad).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double()
    if lrelu:
        y = F.leaky_relu(y, 0.2)
    return y


def ref_dgrad(gy, w, stride, pad, hi, wi):
    k = w.shape[-1]
    op = hi + 2 * pad - k - (gy.shape[1] - 1) * stride
    gx = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double(), None, stride, pad, output_padding=op)
    return gx.permute(0, 2, 3, 1)


def t_us(fn, n=20):
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


def run(n, hi, wi, cin, cout, k, stride, pad, bias=False, res=False, lrelu=False, mask=False, bench=False, check=True, cl=False):
    g = torch.Generator(device='cpu').manual_seed(hi * 131 + cin * 7 + cout + k)
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    x = torch.randn(n, hi, wi, cin, generator=g).to(dev).to(dt)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    if cl:
        w = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)       # channels-last master
    b = torch.randn(cout, generator=g).to(dev) if bias else None
    r = torch.randn(n, ho, wo, cout, generator=g).to(dev).to(dt) if res else None
    gy = torch.randn(n, ho, wo, cout, generator=g).to(dev).to(dt)
    act = ops.ACT_LRELU if lrelu else ops.ACT_NONE
    out = []
    # ---- forward
    pm = oc.plane_mode(dt, n, hi, wi, cin, ho, wo, cout, k, k, stride, pad, False, ops.ACT_NONE, act, ops.AUX_NONE, res)
    tag = 'n%d %dx%d c%d->%d k%d s%d%s%s%s' % (n, hi, wi, cin, cout, k, stride, ' +b' if bias else '', ' +res' if res else '', ' +lrelu' if lrelu else '')
    wb = w.to(dt).float()
    if pm:
        wp = oc.pack_weight(w, dt, cin, False, None, plane=True)
        y = oc.conv2d_raw(x, wp, b, r, None, (ho, wo, cout), k, k, stride, pad, False, ops.ACT_NONE, act, plane=True)
        if check:
            yr = ref_fwd(x, wb, b, stride, pad, r, lrelu)
            err = float((y.double() - yr).abs().max() / yr.abs().max())
            assert err < 1.5e-2, ('F', tag, err)
            out.append('F mode %d rel %.2e' % (pm, err))
        if bench:
            wg = oc.pack_weight(w, dt, cin, False, None)
            tp = t_us(lambda: oc.conv2d_raw(x, wp, b, r, None, (ho, wo, cout), k, k, stride, pad, False, ops.ACT_NONE, act, plane=True))
            tg = t_us(lambda: oc.conv2d_raw(x, wg, b, r, None, (ho, wo, cout), k, k, stride, pad, False, ops.ACT_NONE, act))
            fl = 2.0 * n * ho * wo * cin * cout * k * k
            out.append('F plane %.1f us (%.0f TF) | other %.1f us' % (tp, fl / tp * 1e-6, tg))
    else:
        out.append('F: not a plane shape')
    # ---- data gradient
    aux = x if mask else None
    am = ops.AUX_LRELU_GRAD if mask else ops.AUX_NONE
    pm = oc.plane_mode(dt, n, ho, wo, cout, hi, wi, cin, k, k, stride, pad, True, ops.ACT_NONE, ops.ACT_NONE, am)
    if pm:
        wpt = oc.pack_weight(w, dt, cin, True, None, plane=True)
        gx = oc.conv2d_raw(gy, wpt, None, None, aux, (hi, wi, cin), k, k, stride, pad, True, ops.ACT_NONE, ops.ACT_NONE, am, plane=True)
        if check:
            gr = ref_dgrad(gy, wb, stride, pad, hi, wi)
            if mask:
                gr = gr * torch.where(x.double() > 0, 1.0, 0.2)
            err = float((gx.double() - gr).abs().max() / gr.abs().max())
            assert err < 1.5e-2, ('D', tag, err)
            out.append('D mode %d rel %.2e' % (pm, err))
        if bench:
            wgt = oc.pack_weight(w, dt, cin, True, None)
            tp = t_us(lambda: oc.conv2d_raw(gy, wpt, None, None, aux, (hi, wi, cin), k, k, stride, pad, True, ops.ACT_NONE, ops.ACT_NONE, am, plane=True))
            tg = t_us(lambda: oc.conv2d_raw(gy, wgt, None, None, aux, (hi, wi, cin), k, k, stride, pad, True, ops.ACT_NONE, ops.ACT_NONE, am))
            fl = 2.0 * n * ho * wo * cin * cout * k * k
            out.append('D plane %.1f us (%.0f TF) | other %.1f us' % (tp, fl / tp * 1e-6, tg))
    else:
        out.append('D: not a plane shape')
    print('%-44s %s' % (tag, ' ; '.join(out)), flush=True)


def main():
    bench = '--bench' in sys.argv
    # parity: small shapes, every epilogue, ragged maps, both source layouts
    run(2, 32, 32, 128, 64, 1, 1, 0)
    run(2, 32, 48, 256, 128, 1, 1, 0, res=True, cl=True)
    run(3, 24, 40, 128, 192, 1, 1, 0, bias=True, lrelu=True)
    run(2, 40, 40, 128, 64, 1, 1, 0, mask=True)
    run(4, 64, 64, 64, 128, 3, 2, 1)
    run(4, 64, 64, 64, 128, 3, 2, 1, bias=True, lrelu=True, cl=True)
    run(3, 48, 64, 128, 64, 3, 2, 1, res=True)
    run(2, 32, 32, 256, 512, 3, 2, 1)
    run(5, 36, 44, 64, 64, 3, 2, 1, mask=True)
    run(2, 64, 64, 64, 128, 4, 2, 2)
    run(3, 65, 65, 64, 128, 4, 2, 2, bias=True, lrelu=True, cl=True)
    run(2, 33, 33, 128, 64, 4, 2, 2, mask=True)
    run(2, 129, 129, 64, 128, 4, 2, 2)
    if os.environ.get('S2E_CONV_PLANE') and (int(os.environ['S2E_CONV_PLANE']) >> 2) & 1:
        run(2, 32, 32, 64, 128, 3, 1, 1, bias=True)
        run(2, 24, 40, 128, 64, 3, 1, 1, res=True)
    print('parity ok')
    if bench:
        # netE (batch 32) and the learned shortcuts at the benchmark's sizes
        for (n, h, cin, cout, k, s, p) in [(16, 129, 64, 128, 4, 2, 2), (16, 65, 128, 256, 4, 2, 2), (16, 65, 64, 128, 4, 2, 2), (16, 33, 128, 256, 4, 2, 2),
                                           (32, 128, 64, 128, 3, 2, 1), (32, 64, 128, 256, 3, 2, 1), (32, 32, 256, 512, 3, 2, 1), (32, 16, 512, 512, 3, 2, 1),
                                           (8, 256, 128, 64, 1, 1, 0), (8, 128, 256, 128, 1, 1, 0), (8, 64, 512, 256, 1, 1, 0), (8, 32, 1024, 512, 1, 1, 0)]:
            run(n, h, h, cin, cout, k, s, p, bench=True, check=False)
        if os.environ.get('S2E_CONV_PLANE') and (int(os.environ['S2E_CONV_PLANE']) >> 2) & 1:
            for (n, h, cin, cout) in [(8, 256, 128, 256), (8, 128, 256, 128), (8, 64, 512, 256), (8, 256, 64, 64), (8, 32, 512, 512), (8, 16, 1024, 1024)]:
                run(n, h, h, cin, cout, 3, 1, 1, bench=True, check=False)


if __name__ == '__main__':
    main()
