"""Op-level parity of every HIP kernel (through the C-ABI / ctypes path) against plain
PyTorch fp64 CPU references of the same op.  fp32 kernels: tight tolerance; bf16 kernels:
both sides start from the same bf16-rounded inputs, tolerance = bf16 output rounding."""
import os
import numpy as np
import pytest
from conftest import load_golden
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    return torch.device('cuda:0')


def _rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale).to(dtype)
    return t


def _tol(dtype):
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)


def _close(got, ref, dtype, scale=None, what=''):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    s = max(s, 1e-6)
    tol = 1.5e-2 if dtype == torch.bfloat16 else 1e-4
    err = float((got - ref).abs().max())
    assert err <= tol * s, '%s: max err %.3e vs scale %.3e (tol %.1e)' % (what, err, s, tol)


def _close_kink(got, ref, dtype, what='', frac=5e-4):
    """_close for gradients behind a LeakyReLU whose mask is taken from a bf16 value: an element whose pre-activation lies within
    rounding of zero may take the other slope (its gradient then differs by a factor 5) -- allowed for a `frac` of the elements;
    everything else must agree as in _close, and so must the relative RMS over all elements."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    s_ = max(float(ref.abs().max()), 1e-6)
    tol = 1.5e-2 if dtype == torch.bfloat16 else 1e-4
    bad = ((got - ref).abs() > tol * s_).double().mean()
    rms = float(((got - ref) ** 2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-30))
    assert float(bad) <= frac and rms <= (2e-2 if dtype == torch.bfloat16 else 1e-4), '%s: %.2e of the elements off, rel-RMS %.3e' % (what, float(bad), rms)


def nhwc(t):      # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # (N, H, W, Cin, Cout, k, stride, pad, bias, residual, in_act, out_act, cin_pad)
    (2, 16, 16, 16, 32, 3, 1, 1, True, False, 0, 0, None),
    (2, 16, 12, 64, 128, 3, 1, 1, True, True, 0, 0, None),
    (1, 9, 11, 128, 200, 3, 1, 1, False, False, 0, 0, None),     # ragged M and Cout (masked tiles)
    (2, 16, 16, 32, 16, 1, 1, 0, False, False, 0, 0, None),      # 1x1 shortcut
    (2, 33, 33, 8, 64, 4, 2, 2, True, False, 0, 0, None),        # D model0-like (padded 5->8 elsewhere)
    (2, 17, 17, 64, 128, 4, 2, 2, False, False, 0, 0, None),     # D stride-2
    (2, 9, 9, 32, 64, 4, 1, 2, False, False, 0, 0, None),        # D stride-1 pad 2
    (2, 10, 10, 64, 1, 4, 1, 2, True, False, 0, 0, None),        # D head, Cout=1
    (2, 32, 32, 64, 1, 3, 1, 1, True, False, 1, 2, None),        # conv_img: lrelu -> conv -> tanh
    (3, 32, 32, 1, 16, 3, 2, 1, False, False, 0, 0, None),       # encoder layer0, Cin=1 (gather fallback)
    (2, 32, 32, 16, 32, 3, 2, 1, False, False, 0, 0, None),      # encoder stride-2
    (2, 16, 16, 5, 16, 4, 2, 2, True, False, 0, 0, 8),           # 5 real channels stored as 8
    (2, 8, 8, 256, 256, 3, 1, 1, True, True, 0, 0, None),        # deep K (36 K-tiles)
    (2, 64, 64, 8, 8, 3, 1, 1, True, True, 0, 0, None),          # tiny channels: several taps per K-tile
    (2, 16, 16, 8, 16, 3, 1, 1, True, False, 0, 0, None),
    (2, 16, 16, 16, 8, 1, 1, 0, False, False, 0, 0, None),
    (1, 4, 4, 512, 512, 3, 1, 1, True, True, 0, 0, None),         # split-K: 4 tiles, 72 K-tiles
    (4, 16, 16, 256, 192, 3, 2, 1, False, False, 0, 0, None),    # split-K, stride 2, ragged Cout
    (2, 8, 8, 512, 1, 4, 1, 2, True, False, 0, 0, None),         # split-K with Cout = 1 (scalar finish)
    (2, 8, 8, 128, 256, 3, 1, 1, True, False, 1, 2, None),       # split-K + lrelu prologue + tanh epilogue
    (1, 64, 64, 16, 128, 3, 1, 1, True, True, 0, 0, None),       # BN = 128 tile WITHOUT split-K (3 K-tiles): the big layers' path
    (1, 48, 40, 32, 72, 3, 1, 1, True, False, 0, 0, None),       # ... and the 64 < Cout <= 128 ragged variant, 5 K-tiles
    (3, 15, 22, 24, 40, 4, 2, 2, True, False, 0, 0, None),       # stride-2 dgrad by parity class: odd x even, ragged channels
    (1, 13, 9, 16, 24, 3, 2, 1, False, False, 0, 0, None),       # ... 3x3: classes with 2x2, 2x1, 1x2, 1x1 taps
    (4, 128, 128, 64, 256, 3, 1, 1, True, True, 0, 0, None),     # patch-resident kernels (conv AND wgrad): 64-wide rectangles, BN = 128
    (11, 62, 90, 128, 128, 3, 1, 1, True, False, 0, 0, None),    # ... 32-wide rectangles ragged in y and x; forward, dgrad, wgrad (2 ci tiles)
    (16, 16, 16, 512, 512, 3, 1, 1, True, False, 0, 0, None),    # ... wgrad with 16-wide slabs (8 x 16), 8 ci tiles x 4 co tiles x 8 splits
    (8, 64, 64, 8, 128, 3, 1, 1, True, False, 0, 0, None),       # 8-channel input (label-map convs): wgrad with the B operand built from a 16-B/pixel patch
    (10, 60, 90, 8, 256, 3, 1, 1, True, False, 0, 0, None),      # ... 32-wide slabs, ragged, two co tiles
    (3, 250, 256, 64, 64, 3, 1, 1, True, False, 0, 2, None),     # ... BN = 64 (wgrad: half-empty co tile), ragged in y, tanh epilogue
    (1, 20, 300, 64, 1, 3, 1, 1, True, False, 1, 0, None),       # Cout = 1, wider than one 256-column segment of the band kernels (dgrad, wgrad); dot-then-stencil forward
    (2, 40, 40, 128, 1, 3, 1, 1, True, True, 0, 0, None),        # ... 128 channels (8 K-steps a wave), residual
    (2, 20, 21, 256, 1, 4, 1, 2, False, False, 1, 0, None),      # ... 256 channels: four channel quarters, one LDS slab each
    (1, 12, 600, 1, 16, 3, 2, 1, True, False, 1, 0, None),       # Cin = 1, two column segments, LeakyReLU applied while the patch is staged
    (8, 256, 256, 64, 1, 3, 1, 1, False, False, 1, 2, None),     # the degenerate-channel layers AS BENCHED: the generator's image conv (32 x 32 rectangles),
    (16, 34, 34, 512, 1, 4, 1, 2, False, False, 0, 0, None),     # ... a PatchGAN head (9 x 9 rectangles, four channel quarters),
    (32, 256, 256, 1, 64, 3, 2, 1, False, False, 0, 0, None),    # ... the encoder's first layer (1024 bands of four output rows)
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_forward_backward(case, dtype):
    from seg2eye_amd import ops
    N, H, W, Cin, Cout, k, s, p, has_b, has_r, in_act, out_act, cin_pad = case
    dev = _dev()
    x = _rnd((N, Cin, H, W), 1, dtype)
    w = _rnd((Cout, Cin, k, k), 2, torch.float32, (1.0 / (Cin * k * k)) ** 0.5)
    b = _rnd((Cout,), 3, torch.float32, 0.1) if has_b else None
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    r = _rnd((N, Cout, Ho, Wo), 4, dtype) if has_r else None
    gy = _rnd((N, Cout, Ho, Wo), 5, dtype)

    # reference (fp64 CPU), weights rounded to the compute dtype like the packed ones
    xr = x.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    rr = r.double().requires_grad_(True) if has_r else None
    xi = F.leaky_relu(xr, 0.2) if in_act == 1 else xr
    yr = F.conv2d(xi, wr, br, stride=s, padding=p)
    if has_r:
        yr = yr + rr
    if out_act == 2:
        yr = torch.tanh(yr)
    yr.backward(gy.double())

    xg = nhwc(x)
    if cin_pad:
        xg = torch.cat([xg, torch.zeros(N, H, W, cin_pad - Cin, dtype=dtype)], dim=-1)
    xg = xg.to(dev).requires_grad_(True)
    wg = w.to(dev).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True) if has_b else None
    rg = nhwc(r).to(dev).requires_grad_(True) if has_r else None
    y = ops.conv2d(xg, wg, bg, rg, s, p, in_act, out_act)
    assert y.shape == (N, Ho, Wo, Cout)
    _close(nchw(y), yr, dtype, what='y')
    y.backward(nhwc(gy).to(dev))
    gx = xg.grad[..., :Cin] if cin_pad else xg.grad
    _close(nchw(gx), xr.grad, dtype, what='dx')
    if cin_pad:
        assert float(xg.grad[..., Cin:].abs().max()) == 0.0
    _close(wg.grad, wr.grad, dtype, what='dw')
    if has_b:
        _close(bg.grad, br.grad, dtype, what='db')
    if has_r:
        _close(nchw(rg.grad), rr.grad, dtype, what='dres')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 16, 32), (3, 7, 5, 8), (1, 64, 64, 64), (2, 4, 4, 1024), (2, 8, 8, 48), (2, 34, 34, 72), (1, 36, 36, 16)])
def test_in_stats_and_instance_norm(shape, dtype):
    from seg2eye_amd import ops
    N, H, W, C = shape
    x = _rnd((N, C, H, W), 7, dtype) * 1.5 + 0.7
    gy = _rnd((N, C, H, W), 8, dtype)
    xr = x.double().requires_grad_(True)
    for lrelu in (False, True):
        xr.grad = None
        yr = F.instance_norm(xr, eps=1e-5)
        if lrelu:
            yr = F.leaky_relu(yr, 0.2)
        yr.backward(gy.double())
        xg = nhwc(x).to(_dev()).requires_grad_(True)
        st = ops.in_stats(xg.detach())
        mean = x.double().mean(dim=(2, 3))
        rstd = 1.0 / torch.sqrt(x.double().var(dim=(2, 3), unbiased=False) + 1e-5)
        _close(st[..., 0], mean, torch.float32, what='mean')
        _close(st[..., 1], rstd, torch.float32, what='rstd')
        y = ops.instance_norm(xg, lrelu)
        _close(nchw(y), yr, dtype, what='in out')
        y.backward(nhwc(gy).to(_dev()))
        _close(nchw(xg.grad), xr.grad, dtype, what='in dx')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 16, 32), (2, 9, 7, 8), (1, 32, 32, 128), (2, 64, 64, 8), (2, 64, 64, 16)])
@pytest.mark.parametrize('lrelu', [False, True])
def test_spade_style_modulate(shape, dtype, lrelu):
    from seg2eye_amd import ops
    N, H, W, C = shape
    x = _rnd((N, C, H, W), 11, dtype) * 1.3 + 0.2
    gb = _rnd((N, 2 * C, H, W), 12, dtype, 0.5)
    style = _rnd((N, 2 * C), 13, torch.float32, 0.5)
    gy = _rnd((N, C, H, W), 14, dtype)
    xr, gbr, sr = x.double().requires_grad_(True), gb.double().requires_grad_(True), style.double().requires_grad_(True)
    gamma, beta = gbr[:, :C], gbr[:, C:]
    s0, s1 = sr[:, :C, None, None], sr[:, C:, None, None]
    yr = 0.5 * (F.instance_norm(xr, eps=1e-5) * (1 + gamma) + beta + xr * (1 + s0) + s1)
    if lrelu:
        yr = F.leaky_relu(yr, 0.2)
    yr.backward(gy.double())
    dev = _dev()
    xg = nhwc(x).to(dev).requires_grad_(True)
    gbg = nhwc(gb).to(dev).requires_grad_(True)
    sg = style.to(dev).requires_grad_(True)
    y = ops.spade_style_modulate(xg, gbg, sg, ops.in_stats(xg.detach()), lrelu)
    _close(nchw(y), yr, dtype, what='mod out')
    y.backward(nhwc(gy).to(dev))
    _close(nchw(xg.grad), xr.grad, dtype, what='mod dx')
    _close(nchw(gbg.grad), gbr.grad, dtype, what='mod dgb')
    _close(sg.grad, sr.grad, dtype, scale=float(sr.grad.abs().max()), what='mod dstyle')


def _labels(n, H, W, seed):
    from seg2eye_amd.synthetic import ellipse_labels
    lab = ellipse_labels(n, H, W, seed)[:, 0]
    g = np.random.RandomState(seed)
    noise = g.randint(0, 4, size=lab.shape).astype(np.uint8)
    m = g.rand(*lab.shape) < 0.15                       # salt some random classes in
    lab = np.where(m, noise, lab).astype(np.uint8)
    return torch.from_numpy(lab)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(2, 32, 32, 32, 32, 128), (2, 64, 32, 8, 4, 128), (2, 64, 64, 2, 2, 256), (1, 32, 32, 16, 16, 24)])
def test_spade_params_and_label_conv(cfg, dtype):
    from seg2eye_amd import ops
    N, H, W, h, w, C2 = cfg
    dev = _dev()
    lab = _labels(N, H, W, 3)
    onehot = torch.zeros(N, 4, H, W).scatter_(1, lab.long().unsqueeze(1), 1.0)
    seg_h = F.interpolate(onehot, size=(h, w), mode='nearest').double()
    w_sh = _rnd((128, 4, 3, 3), 21, torch.float32, 0.3)
    b_sh = _rnd((128,), 22, torch.float32, 0.1)
    w_gb = _rnd((C2, 128, 3, 3), 23, torch.float32, 0.03)
    b_gb = _rnd((C2,), 24, torch.float32, 0.1)
    Ch = C2 // 2
    ggb = _rnd((N, C2, h, w), 25, dtype)
    refs = [t.double().requires_grad_(True) for t in (w_sh, b_sh, w_gb.to(dtype), b_gb)]
    actv = F.relu(F.conv2d(seg_h, refs[0], refs[1], padding=1))
    if dtype == torch.bfloat16:
        actv = actv + (actv.detach().to(dtype).double() - actv.detach())      # the kernel stores actv in bf16
    gbr = F.conv2d(actv, refs[2], refs[3], padding=1)
    gbr.backward(ggb.double())
    # (a) separate gamma / beta parameters -> concatenated inside, gradients through autograd
    prm = [t.to(dev).requires_grad_(True) for t in (w_sh, b_sh, w_gb[:Ch].clone(), b_gb[:Ch].clone(), w_gb[Ch:].clone(), b_gb[Ch:].clone())]
    gb = ops.spade_params(lab.to(dev), *prm, h, w, dtype)
    _close(nchw(gb), gbr, dtype, what='gb')
    gb.backward(nhwc(ggb).to(dev))
    _close(prm[0].grad, refs[0].grad, dtype, what='dw_sh')
    _close(prm[1].grad, refs[1].grad, dtype, what='db_sh')
    _close(torch.cat([prm[2].grad, prm[4].grad]), refs[2].grad, dtype, what='dw_gb')
    _close(torch.cat([prm[3].grad, prm[5].grad]), refs[3].grad, dtype, what='db_gb')
    # (b) parameters living in a flat arena (gamma/beta adjacent): zero-copy [gamma|beta], direct .grad accumulation
    from seg2eye_amd.optim import FlatAdam
    prm2 = [torch.nn.Parameter(t.to(dev)) for t in (w_sh, b_sh, w_gb[:Ch].clone(), w_gb[Ch:].clone(), b_gb[:Ch].clone(), b_gb[Ch:].clone())]
    fa = FlatAdam(prm2, lr=1e-3)
    for _ in range(2):                                   # twice: accumulation, not overwrite
        gb2 = ops.spade_params(lab.to(dev), prm2[0], prm2[1], prm2[2], prm2[4], prm2[3], prm2[5], h, w, dtype)
        gb2.backward(nhwc(ggb).to(dev))
    _close(nchw(gb2), gbr, dtype, what='gb (arena)')
    _close(prm2[0].grad, 2 * refs[0].grad, dtype, what='dw_sh (arena)')
    _close(prm2[1].grad, 2 * refs[1].grad, dtype, what='db_sh (arena)')
    _close(torch.cat([prm2[2].grad, prm2[3].grad]), 2 * refs[2].grad, dtype, what='dw_gb (arena)')
    _close(torch.cat([prm2[4].grad, prm2[5].grad]), 2 * refs[3].grad, dtype, what='db_gb (arena)')
    assert prm2[2].grad.data_ptr() >= fa.flat_g.data_ptr()
    # plain label conv (generator fc), no ReLU, wide Cout
    w_fc = _rnd((C2 * 2, 4, 3, 3), 26, torch.float32, 0.3)
    b_fc = _rnd((C2 * 2,), 27, torch.float32, 0.1)
    gfc = _rnd((N, C2 * 2, h, w), 28, dtype)
    wr, br = w_fc.double().requires_grad_(True), b_fc.double().requires_grad_(True)
    yr = F.conv2d(seg_h, wr, br, padding=1)
    yr.backward(gfc.double())
    wg, bg = w_fc.to(dev).requires_grad_(True), b_fc.to(dev).requires_grad_(True)
    y = ops.label_conv3x3(lab.to(dev), wg, bg, h, w, False, dtype)
    _close(nchw(y), yr, dtype, what='fc')
    y.backward(nhwc(gfc).to(dev))
    _close(wg.grad, wr.grad, dtype, what='fc dw')
    _close(bg.grad, br.grad, dtype, what='fc db')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('sparse', [False, True])
@pytest.mark.parametrize('cfg', [(2, 64, 64, 32, 32, 64, True, True), (1, 48, 80, 24, 40, 128, False, True), (2, 32, 32, 16, 16, 192, True, False),
                                 (3, 64, 64, 64, 64, 64, True, True), (2, 128, 128, 128, 128, 64, True, False)])
def test_spade_conv_modulate_fused(cfg, sparse, dtype):
    """s2e_spade_conv_modulate (the [gamma | beta] conv with the modulation in its epilogue; flags=1 forces the fused kernel at
    any tile count) against the fp64 reference of SPADE_STYLE_Block.forward (normalization.py:184-192) and against the two-launch
    path, forward (no-grad: gamma never stored; with grad) and every gradient (s2e_modulate_bwd_gamma).
    sparse: the label-sparse form (flags 4 forces it at these small sizes) on clean nested-ellipse maps -- label-uniform
    rectangles take gamma / beta from the per-class table, incl. the rectangles on the image border; dense: flags 2."""
    from seg2eye_amd import ops
    from seg2eye_amd.synthetic import ellipse_labels
    N, H, W, h, w, C, lrelu, relay = cfg
    dev = _dev()
    lab = torch.from_numpy(ellipse_labels(N, H, W, 5)[:, 0]) if sparse else _labels(N, H, W, 5)
    FL = 1 | (4 if sparse else 2)
    if sparse:                                           # the map must really have both kinds of rectangles
        rc = ops.label_rects(lab.to(dev), h, w, dtype, C, 128, FL)
        assert rc is not None
        nd, nu = int(rc[3][0]), int(rc[3][1])
        assert nd + nu == rc[0].numel() and nd > 0 and (nu > 0 or h < 64), (nd, nu)     # small maps: every rectangle crosses a boundary
    onehot = torch.zeros(N, 4, H, W).scatter_(1, lab.long().unsqueeze(1), 1.0)
    seg_h = F.interpolate(onehot, size=(h, w), mode='nearest').double()
    x = _rnd((N, C, h, w), 41, dtype) * 1.3 + 0.2
    style = _rnd((N, 2 * C), 42, torch.float32, 0.5)
    gy = _rnd((N, C, h, w), 43, dtype)
    w_sh = _rnd((128, 4, 3, 3), 44, torch.float32, 0.3)
    b_sh = _rnd((128,), 45, torch.float32, 0.1)
    w_gb = _rnd((2 * C, 128, 3, 3), 46, torch.float32, 0.03)
    b_gb = _rnd((2 * C,), 47, torch.float32, 0.1)
    refs = [t.double().requires_grad_(True) for t in (w_sh, b_sh, w_gb.to(dtype), b_gb)]
    xr, sr = x.double().requires_grad_(True), style.double().requires_grad_(True)
    actv = F.relu(F.conv2d(seg_h, refs[0], refs[1], padding=1))
    if dtype == torch.bfloat16:
        actv = actv + (actv.detach().to(dtype).double() - actv.detach())      # the kernel stores actv in bf16
    gbr = F.conv2d(actv, refs[2], refs[3], padding=1)
    gamma, beta = gbr[:, :C], gbr[:, C:]
    s0, s1 = sr[:, :C, None, None], sr[:, C:, None, None]
    yr = 0.5 * (F.instance_norm(xr, eps=1e-5) * (1 + gamma) + beta + xr * (1 + s0) + s1)
    if lrelu:
        yr = F.leaky_relu(yr, 0.2)
    yr.backward(gy.double())
    assert ops.spade_fused_supported(nhwc(x).to(dev), 128, flags=1)
    prm = [t.to(dev).requires_grad_(True) for t in (w_sh, b_sh, w_gb[:C].clone(), b_gb[:C].clone(), w_gb[C:].clone(), b_gb[C:].clone())]
    xg = nhwc(x).to(dev).requires_grad_(True)
    sg = style.to(dev).requires_grad_(True)
    st = ops.in_stats(xg.detach())
    with torch.no_grad():
        y0 = ops.spade_style_fused(xg, lab.to(dev), *prm, sg, st, lrelu, flags=FL)
    _close(nchw(y0), yr, dtype, what='fused out (no grad)')
    # flags 8: x handed over at HALF resolution, the generator's nearest 2x upsampling folded into the launch's read of x
    # (no-grad forward) -- the same bits as upsampling first
    xl = nhwc(_rnd((N, C, h // 2, w // 2), 48, dtype) * 1.1 - 0.1).to(dev)
    stl = ops.in_stats(xl)
    with torch.no_grad():
        y_fold = ops.spade_style_fused(xl, lab.to(dev), *prm, sg, stl, lrelu, flags=FL | 8)
        y_mat = ops.spade_style_fused(ops.upsample2x(xl), lab.to(dev), *prm, sg, stl, lrelu, flags=FL)
    assert y_fold.shape == y_mat.shape == (N, h, w, C) and torch.equal(y_fold, y_mat)
    # ... and WITH gradients (the training forward): the same output bit for bit, and the gradient w.r.t. the half-resolution
    # tensor -- the launch's backward sums the 2 x 2 pixels in fp32 and rounds once (s2e_modulate_bwd_staged: x_up_w, dx_quad),
    # upsampling first rounds four times and once more in the upsampling's backward: equal to the compute dtype's resolution
    gyl = nhwc(gy).to(dev)
    res = []
    for fold in (False, True):
        xs = xl.clone().requires_grad_(True)
        prm2 = [t.detach().clone().requires_grad_(True) for t in prm]
        if fold:
            yy = ops.spade_style_fused(xs, lab.to(dev), *prm2, sg.detach(), stl, lrelu, flags=FL | 8)
        else:
            yy = ops.spade_style_fused(ops.upsample2x(xs), lab.to(dev), *prm2, sg.detach(), stl, lrelu, flags=FL)
        yy.backward(gyl)
        res.append((yy.detach(), xs.grad, [t.grad for t in prm2]))
    assert torch.equal(res[0][0], res[1][0])
    _close(res[1][1], res[0][1], dtype, what='dx of the folded upsampling')
    for a, b in zip(res[0][2], res[1][2]):
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-6      # (weight-gradient atomics: summation order)
    y = ops.spade_style_fused(xg, lab.to(dev), *prm, sg, st, lrelu, relay=relay, flags=FL)
    if relay:
        y, xalias = y
        assert xalias.data_ptr() == xg.data_ptr()
    assert torch.equal(y, y0)
    y.backward(nhwc(gy).to(dev))
    # (sparse + bf16: gamma / beta of the label-uniform rectangles come from a bf16 table, the LeakyReLU mask from the bf16 output:
    # a few pre-activations within rounding of zero take the other slope than the fp64 reference's)
    (_close_kink if (sparse and dtype == torch.bfloat16) else _close)(nchw(xg.grad), xr.grad, dtype, what='fused dx')
    _close(sg.grad, sr.grad, dtype, scale=float(sr.grad.abs().max()), what='fused dstyle')
    _close(prm[0].grad, refs[0].grad, dtype, what='fused dw_sh')
    _close(prm[1].grad, refs[1].grad, dtype, what='fused db_sh')
    _close(torch.cat([prm[2].grad, prm[4].grad]), refs[2].grad, dtype, what='fused dw_gb')
    _close(torch.cat([prm[3].grad, prm[5].grad]), refs[3].grad, dtype, what='fused db_gb')
    # the two-launch path on the same inputs: same result up to the bf16 rounding of gamma / beta it stores in between
    gb2 = ops.spade_params(lab.to(dev), *[t.detach() for t in prm], h, w, dtype)
    y2 = ops.spade_style_modulate(xg.detach(), gb2, sg.detach(), st, lrelu)
    _close(y, y2, dtype, what='fused vs two-launch')


@pytest.mark.parametrize('cfg', [(4, 128, 128, 64), (2, 256, 256, 128), (4, 128, 256, 256)])
def test_spade_label_sparse_backward(cfg):
    """The label-sparse BACKWARD of the SPADE branch (csrc/spade_sparse_bwd.hip): with parameters in an
    optimizer arena and a trainer-step scope open, the [gamma | beta] conv's data gradient runs on the rectangles that cross a label
    boundary (or touch the border) only, and the uniform-interior ones reach mlp_shared's gradients through nine shifted sums of
    d[gamma | beta] per class.  Every parameter gradient against the fp64 reference of normalization.py:91-105 differentiated, with
    the sparse backward on and off (the two must also agree with each other), on nested-ellipse label maps."""
    from seg2eye_amd import ops
    from seg2eye_amd.optim import FlatAdam
    from seg2eye_amd.synthetic import ellipse_labels
    N, H, W, C = cfg
    dev, dtype = _dev(), torch.bfloat16
    lab = torch.from_numpy(ellipse_labels(N, H, W, 7)[:, 0])
    onehot = torch.zeros(N, 4, H, W).scatter_(1, lab.long().unsqueeze(1), 1.0).double()
    x = _rnd((N, C, H, W), 51, dtype) * 1.3 + 0.2
    style = _rnd((N, 2 * C), 52, torch.float32, 0.5)
    gy = _rnd((N, C, H, W), 53, dtype)
    w_sh = _rnd((128, 4, 3, 3), 54, torch.float32, 0.3)
    b_sh = _rnd((128,), 55, torch.float32, 0.1)
    w_gb = _rnd((2 * C, 128, 3, 3), 56, torch.float32, 0.03)
    b_gb = _rnd((2 * C,), 57, torch.float32, 0.1)
    refs = [t.double().requires_grad_(True) for t in (w_sh, b_sh, w_gb.to(dtype), b_gb)]
    xr = x.double()
    actv = F.relu(F.conv2d(onehot, refs[0], refs[1], padding=1))
    actv = actv + (actv.detach().to(dtype).double() - actv.detach())          # the kernel stores actv in bf16
    gbr = F.conv2d(actv, refs[2], refs[3], padding=1)
    s0, s1 = style.double()[:, :C, None, None], style.double()[:, C:, None, None]
    yr = F.leaky_relu(0.5 * (F.instance_norm(xr, eps=1e-5) * (1 + gbr[:, :C]) + gbr[:, C:] + xr * (1 + s0) + s1), 0.2)
    yr.backward(gy.double())
    got = {}
    for off in (True, False):
        ops.switches.SPARSE_BWD_OFF = off
        try:
            prm = [torch.nn.Parameter(t.to(dev)) for t in (w_sh, b_sh, w_gb[:C].clone(), b_gb[:C].clone(), w_gb[C:].clone(), b_gb[C:].clone())]
            fa = FlatAdam([prm[0], prm[1], prm[2], prm[4], prm[3], prm[5]], lr=1e-3)         # gamma / beta weights (and biases) adjacent
            fa.zero_grad()
            xg = nhwc(x).to(dev).requires_grad_(True)
            st = ops.in_stats(xg.detach())
            pool = ops.ZeroPool(dev)
            with pool.scope('t'):
                y = ops.spade_style_fused(xg, lab.to(dev), *prm, style.to(dev), st, True)
                rc = ops.label_rects(lab.to(dev), H, W, dtype, C, 128, 0)
                assert rc is not None and rc[4] == 16 and rc[5] == 16
                y.backward(nhwc(gy).to(dev))
                queued = len(pool.sink.uni)
            assert queued == (0 if off else 1), queued                             # the sparse backward really ran (or really did not)
            if not off:
                sp = pool.step_cache                                              # (cleared at scope exit: counts were read inside)
            got[off] = [q.grad.detach().clone() for q in prm]
        finally:
            ops.switches.SPARSE_BWD_OFF = False
    for off in (True, False):
        g = got[off]
        tag = 'dense' if off else 'sparse'
        _close(g[0], refs[0].grad, dtype, what='dw_sh (%s backward)' % tag)
        _close(g[1], refs[1].grad, dtype, what='db_sh (%s backward)' % tag)
        _close(torch.cat([g[2], g[4]]), refs[2].grad, dtype, what='dw_gb (%s backward)' % tag)
        _close(torch.cat([g[3], g[5]]), refs[3].grad, dtype, what='db_gb (%s backward)' % tag)
    for a, b in zip(got[True], got[False]):
        assert float((a - b).abs().max()) <= 6e-3 * float(a.abs().max()) + 1e-6, float((a - b).abs().max()) / float(a.abs().max())


@pytest.mark.parametrize('dtype', DTYPES)
def test_resampling_and_concat(dtype):
    from seg2eye_amd import ops
    dev = _dev()
    x = _rnd((2, 16, 5, 7), 31, dtype)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode='nearest')
    gy = _rnd(tuple(yr.shape), 32, dtype)
    yr.backward(gy.double())
    xg = nhwc(x).to(dev).requires_grad_(True)
    y = ops.upsample2x(xg)
    assert torch.equal(nchw(y).cpu().double(), yr.detach())
    y.backward(nhwc(gy).to(dev))
    _close(nchw(xg.grad), xr.grad, dtype, what='ups bwd')
    for H, W in ((32, 32), (17, 13)):
        x = _rnd((3, 8, H, W), 33, dtype)
        xr = x.double().requires_grad_(True)
        yr = F.avg_pool2d(xr, 3, stride=2, padding=[1, 1], count_include_pad=False)
        gy = _rnd(tuple(yr.shape), 34, dtype)
        yr.backward(gy.double())
        xg = nhwc(x).to(dev).requires_grad_(True)
        y = ops.avgpool3x3s2(xg)
        _close(nchw(y), yr, dtype, what='pool')
        y.backward(nhwc(gy).to(dev))
        _close(nchw(xg.grad), xr.grad, dtype, what='pool bwd')
    lab = _labels(2, 16, 16, 5)
    img = _rnd((2, 16, 16), 35, dtype)
    ig = img.to(dev).requires_grad_(True)
    cat = ops.seg_image_concat(lab.to(dev), ig)
    ref = torch.cat([torch.zeros(2, 4, 16, 16).scatter_(1, lab.long().unsqueeze(1), 1.0), img.float().unsqueeze(1),
                     torch.zeros(2, 3, 16, 16)], 1)
    assert torch.equal(nchw(cat).float().cpu(), ref)
    g = _rnd((2, 16, 16, 8), 36, dtype).to(dev)
    cat.backward(g)
    assert torch.equal(ig.grad, g[..., 4])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n', [121, 4096, 50003])
def test_losses(dtype, n):
    from seg2eye_amd import ops, _lib as L
    dev = _dev()
    a = _rnd((n,), 41, dtype, 1.5)
    b = _rnd((n,), 42, dtype, 1.5)
    for mode, fn in ((L.LOSS_NEG_MEAN, lambda a, b: -a.sum()),
                     (L.LOSS_HINGE_REAL, lambda a, b: -torch.min(a - 1, torch.zeros_like(a)).sum()),
                     (L.LOSS_HINGE_FAKE, lambda a, b: -torch.min(-a - 1, torch.zeros_like(a)).sum()),
                     (L.LOSS_L1, lambda a, b: (a - b).abs().sum())):
        ar = a.double().requires_grad_(True)
        lr = fn(ar, b.double()) * (0.37 / n)
        (lr * 1.7).backward()
        ag = a.to(dev).requires_grad_(True)
        bg = b.to(dev) if mode == L.LOSS_L1 else None
        l = ops.loss_sum(ag, bg, mode, 0.37 / n)
        assert abs(float(l) - float(lr)) <= 2e-4 * max(1.0, abs(float(lr)))
        (l * 1.7).backward()
        _close(ag.grad, ar.grad, dtype, what='loss grad %d' % mode)


@pytest.mark.parametrize('beta1', [0.0, 0.5])
@pytest.mark.parametrize('wd', [0.0, 0.05])
def test_adam_flat_matches_torch(wd, beta1):
    """s2e_adam_flat against torch.optim.Adam: parameters AND both moments after four steps.  beta1 = 0, wd = 0 is the reference's
    TTUR setting (pix2pix_model.py:98-108): there the kernel skips the first moment altogether (m_t = g_t exactly) and
    FlatAdam.state_dict forms it from the gradient arena."""
    from seg2eye_amd.optim import FlatAdam
    dev = _dev()
    n = 10007
    p0 = _rnd((n,), 51, torch.float32)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(beta1, 0.9), eps=1e-8, weight_decay=wd)
    mine = torch.nn.Parameter(p0.to(dev).clone())
    fa = FlatAdam([mine], lr=1e-3, betas=(beta1, 0.9), weight_decay=wd)
    for step in range(1, 5):
        g = _rnd((n,), 60 + step, torch.float32)
        if step == 3:
            fa.param_groups[0]['lr'] = 5e-4              # LR decay reaches the device-side hyper block
            opt.param_groups[0]['lr'] = 5e-4
        ref.grad = g.clone()
        opt.step()
        mine.grad.copy_(g.to(dev) * 2.0)                 # a 2-rank sum all-reduce ...
        fa.step(grad_scale=0.5)                          # ... averaged inside the kernel
    assert float(fa.hyper[4]) == 4.0
    np.testing.assert_allclose(mine.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    sd, st = fa.state_dict(), opt.state[ref]
    np.testing.assert_allclose(sd['v'][:n].cpu().numpy(), st['exp_avg_sq'].numpy(), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(sd['m'][:n].cpu().numpy(), st['exp_avg'].numpy(), rtol=1e-5, atol=1e-7)
    if beta1 == 0.0 and wd == 0.0:
        assert float(fa.flat_m.abs().max()) == 0.0      # never touched by the kernel


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('arena', [None, 'channels_last', 'torch_order'])
@pytest.mark.parametrize('cfg', [(16, 24, 3, 1, 1), (64, 130, 4, 2, 2), (8, 8, 1, 1, 0), (1, 16, 3, 2, 1)])     # (last: the encoder's 1-channel first layer)
def test_fused_spectral_norm_conv_matches_torch(cfg, arena, dtype):
    """SpectralBank (batched power iteration) + sigma-in-the-pack-kernel + fused gradient through sigma
    against torch.nn.utils.spectral_norm in fp64: train-mode forward (u, v updated), gradient w.r.t.
    weight_orig, a second forward, eval-mode forward, and a 3-iteration step."""
    import copy
    from seg2eye_amd import ops
    from seg2eye_amd.spectral import sn_begin
    cin, cout, k, s, p = cfg
    dev = _dev()
    torch.manual_seed(7)
    ref = torch.nn.utils.spectral_norm(torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=True)).double()
    mine = torch.nn.Sequential(copy.deepcopy(ref).float()).to(dev)
    conv = mine[0]
    if arena is not None:
        # the parameters in an optimizer arena: weight_orig stored channels-last (the weight gradient lands in .grad directly,
        # spectral norm's chain rule runs in place, the power iteration reads W's columns in (tap, ci) order while weight_v
        # keeps torch's order) or in torch's order (re-layout kernels)
        from seg2eye_amd.optim import FlatAdam
        fa = FlatAdam(list(mine.parameters()), lr=1e-3, channels_last=arena == 'channels_last')
        assert conv.weight_orig.is_contiguous() == (arena != 'channels_last' or k == 1 or cin % 8 != 0)
        assert conv.weight_orig.grad.data_ptr() >= fa.flat_g.data_ptr()
    x = _rnd((2, cin, 12, 12), 3, dtype)
    xr = x.double().requires_grad_(True)
    xg = nhwc(x).to(dev).requires_grad_(True)
    for rnd, train in enumerate([True, True, False]):
        ref.train(train)
        mine.train(train)
        ref.zero_grad()
        yr = ref(xr)
        gy = _rnd(tuple(yr.shape), 10 + rnd, dtype)
        yr.backward(gy.double())
        if arena is None:
            conv.zero_grad()
        else:
            fa.zero_grad()                                   # (keeps the .grad views of the arena)
        sn_begin(mine)
        if arena == 'channels_last':
            # as inside a trainer step: the weight gradient lands in the arena slice and the chain rule rewrites it in place
            # (s2e_sn_grads_inplace) when the scope's sink flushes; outside a step the accumulate path runs (test below)
            pool = ops.ZeroPool(dev)
            with pool.scope('t'):
                y = ops.conv2d_m(xg, conv, None, s, p)
                y.backward(nhwc(gy).to(dev))
                if cin % 8 == 0:
                    assert len(pool.sink.inplace) == 1
        else:
            y = ops.conv2d_m(xg, conv, None, s, p)
            y.backward(nhwc(gy).to(dev))
        _close(nchw(y), yr, dtype, what='sn y (round %d)' % rnd)
        _close(conv.weight_orig.grad, ref.weight_orig.grad, dtype, what='sn dW_orig (round %d)' % rnd)
        _close(conv.bias.grad, ref.bias.grad, dtype, what='sn db')
        _close(conv.weight_u, ref.weight_u, torch.float32, what='u (round %d)' % rnd)
        _close(conv.weight_v, ref.weight_v, torch.float32, what='v (round %d)' % rnd)
    # 3 iterations in one step == 3 train-mode forwards of the hook
    ref.train()
    mine.train()
    with torch.no_grad():
        for _ in range(3):
            ref(xr)
        sn_begin(mine, iterations=3)
    _close(conv.weight_u, ref.weight_u, torch.float32, what='u (3 iters)')
    _close(conv.weight_v, ref.weight_v, torch.float32, what='v (3 iters)')
    sd = mine.state_dict()
    assert set(sd) == {'0.bias', '0.weight_orig', '0.weight_u', '0.weight_v'}


@pytest.mark.parametrize('cfg', [(16, 24, 1, 1, 0), (16, 24, 3, 1, 1), (64, 72, 3, 1, 1)])
@pytest.mark.parametrize('where', ['plain_grad', 'arena', 'arena_in_step', 'arena_in_step_twice', 'arena_flushed_between'])
def test_spectral_norm_weight_gradient_accumulates(cfg, where):
    """a spectral-normed conv's weight gradient must ACCUMULATE like torch's whenever .grad is not fresh --
    backward twice without zero_grad (gradient accumulation), a layer applied twice in one step, a second backward behind a
    flush -- on a contiguous 1x1 weight and on channels-last 3x3 weights, with a plain .grad tensor and with an optimizer arena.
    The in-place chain rule (g <- g/sigma - <g, W>/sigma^2 u v^T on the arena slice) may only stand in when the slice was zero
    at the start of the step and the rule runs once over the summed raw gradient; everything else takes the accumulate path."""
    import copy
    from seg2eye_amd import ops
    from seg2eye_amd.optim import FlatAdam
    from seg2eye_amd.spectral import sn_begin
    cin, cout, k, s, p = cfg
    dev = _dev()
    torch.manual_seed(11)
    ref = torch.nn.utils.spectral_norm(torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, bias=False)).double()
    mine = torch.nn.Sequential(copy.deepcopy(ref).float()).to(dev)
    conv = mine[0]
    fa = None
    if where == 'plain_grad':
        conv.weight_orig.grad = torch.zeros_like(conv.weight_orig, memory_format=torch.channels_last)
    else:
        fa = FlatAdam(list(mine.parameters()), lr=1e-3, channels_last=True)
    xs = [_rnd((2, cin, 10, 10), 20 + i, torch.float32) for i in range(2)]
    gys = None
    ref.train(); mine.train()
    ref.zero_grad()
    # reference: ONE power iteration (one forward in train mode), both inputs through the same normalised weight, gradients summed
    ref(xs[0].double())                                       # advances u, v once
    ref.eval()                                                # (further forwards reuse u, v: what sn_begin once + two convs does)
    outs = [ref(x.double()) for x in xs]
    gys = [_rnd(tuple(o.shape), 30 + i, torch.float32) for i, o in enumerate(outs)]
    for o, g in zip(outs, gys):
        o.backward(g.double())
    want = ref.weight_orig.grad
    pool = ops.ZeroPool(dev)
    if fa is not None:
        fa.zero_grad()
    sn_begin(mine)
    xg = [nhwc(x).to(dev).requires_grad_(True) for x in xs]
    gg = [nhwc(g).to(dev) for g in gys]
    if where in ('plain_grad', 'arena'):
        for x, g in zip(xg, gg):                              # two backwards, no zero_grad between, no trainer step around them
            ops.conv2d_m(x, conv, None, s, p).backward(g)
    elif where == 'arena_in_step':
        with pool.scope('t'):                                 # one trainer step: the layer applied twice, one backward
            y0, y1 = ops.conv2d_m(xg[0], conv, None, s, p), ops.conv2d_m(xg[1], conv, None, s, p)
            torch.autograd.backward([y0, y1], gg)
            assert len(pool.sink.inplace) <= 1                # (queued once: both raw contributions are in .grad, the rule is linear)
    elif where == 'arena_in_step_twice':
        with pool.scope('t'):                                 # two backwards inside one step
            for x, g in zip(xg, gg):
                ops.conv2d_m(x, conv, None, s, p).backward(g)
    else:
        with pool.scope('t'):                                 # a flush between the two backwards: the second must not rewrite
            ops.conv2d_m(xg[0], conv, None, s, p).backward(gg[0])
            pool.sink.flush()
            ops.conv2d_m(xg[1], conv, None, s, p).backward(gg[1])
    _close(conv.weight_orig.grad, want, torch.float32, what='accumulated sn dW_orig (%s)' % where)
    if fa is not None:                                        # a second step WITHOUT zero_grad: the arena is not fresh any more
        with pool.scope('t'):
            ops.conv2d_m(xg[0], conv, None, s, p).backward(gg[0])
        ref.weight_orig.grad = None
        outs = ref(xs[0].double())
        outs.backward(gys[0].double())
        _close(conv.weight_orig.grad, want + ref.weight_orig.grad, torch.float32, what='stale arena (%s)' % where)


def test_degenerate_channel_convs_random_shapes():
    """csrc/conv_small.hip (round 5): 40 random 1-channel layers -- Cout = 1 (dot-then-stencil on the matrix cores for 64 / 128 /
    256 / 512 input channels, the vector kernel for the others; data gradient with the taps as one MFMA K-step for whole 32-channel
    blocks, the band kernel otherwise; weight gradient from an LDS copy of the gradient map) and Cin = 1 (the same, mirrored) --
    3x3 and 4x4, any pad, stride 1 or 2 where the kind allows it, odd sizes, widths past one 256-column segment, bias / residual /
    LeakyReLU-in / tanh-out, both dtypes; forward and all three gradients through ops.conv2d against fp64."""
    from seg2eye_amd import ops
    dev = _dev()
    rng = np.random.RandomState(5)
    for it in range(40):
        dtype = torch.bfloat16 if it % 4 else torch.float32
        vec = 8 if dtype == torch.bfloat16 else 4
        k = int(rng.choice([3, 4]))
        wide = int(rng.choice([8, 16, 32, 64, 128, 256, 512] if dtype == torch.bfloat16 else [4, 8, 16, 64, 128, 256]))
        cout1 = bool(rng.randint(2))
        Cin, Cout = (wide, 1) if cout1 else (1, wide)
        s = 1 if cout1 else int(rng.choice([1, 2]))
        p = int(rng.randint(0, k))
        N = int(rng.randint(1, 4))
        H = int(rng.randint(k + 1, 40))
        W = int(rng.choice([rng.randint(k + 1, 40), rng.randint(257, 300) * s])) if it % 5 == 0 else int(rng.randint(k + 1, 40))
        if wide >= 256:
            H, W = min(H, 20), min(W, 24)
        in_act = int(rng.choice([0, 1]))
        out_act = int(rng.choice([0, 0, 2]))
        has_b, has_r = bool(rng.randint(2)), bool(rng.randint(3) == 0)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = _rnd((N, Cin, H, W), 100 + it, dtype)
        w = _rnd((Cout, Cin, k, k), 200 + it, torch.float32, (1.0 / (Cin * k * k)) ** 0.5)
        b = _rnd((Cout,), 300 + it, torch.float32, 0.1) if has_b else None
        r = _rnd((N, Cout, Ho, Wo), 400 + it, dtype) if has_r else None
        gy = _rnd((N, Cout, Ho, Wo), 500 + it, dtype, 1.0 / max(1.0, (N * Ho * Wo) ** 0.5))     # (keeps the bias sum well-conditioned)
        xr = x.double().requires_grad_(True)
        wr = w.to(dtype).double().requires_grad_(True)
        br = b.double().requires_grad_(True) if has_b else None
        yr = F.conv2d(F.leaky_relu(xr, 0.2) if in_act else xr, wr, br, stride=s, padding=p)
        if has_r:
            yr = yr + r.double()
        if out_act == 2:
            yr = torch.tanh(yr)
        yr.backward(gy.double())
        xg = nhwc(x).to(dev).requires_grad_(True)
        wg = w.to(dev).requires_grad_(True)
        bg = b.to(dev).requires_grad_(True) if has_b else None
        rg = nhwc(r).to(dev) if has_r else None
        y = ops.conv2d(xg, wg, bg, rg, s, p, in_act, out_act)
        y.backward(nhwc(gy).to(dev))
        tag = 'case %d: %s N%d %dx%d c%d->%d k%d s%d p%d in%d out%d b%d r%d' % (it, dtype, N, H, W, Cin, Cout, k, s, p, in_act, out_act, has_b, has_r)
        _close(nchw(y), yr, dtype, what=tag + ' y')
        (_close_kink if in_act else _close)(nchw(xg.grad), xr.grad, dtype, what=tag + ' dx')
        _close(wg.grad, wr.grad, dtype, what=tag + ' dw')
        if has_b:
            _close(bg.grad, br.grad, dtype, scale=max(float(br.grad.abs().max()), float(gy.double().abs().sum()) * 1e-2), what=tag + ' db')


@pytest.mark.parametrize('forced', [True, False])
def test_conv2d_random_shapes(forced):
    """60 random 3x3 stride-1 shapes (odd sizes, ragged channels, 8-channel inputs, bias / residual / activations, both
    dtypes) through ops.conv2d forward + backward vs fp64 -- with the patch-resident kernels forced on for every eligible
    shape (thresholds lowered to 1 in a child process) and with the shipped thresholds."""
    import os, subprocess, sys
    env = dict(os.environ, SEED='11' if forced else '12', N='30')
    if forced:
        env.update(S2E_CONV_PATCH='1', S2E_WGRAD_PATCH='1')
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_stress_conv.py')
    out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'done, mismatches: 0' in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize('dtype', DTYPES)
def test_openeds_metric_kernels(dtype):
    """SURVEY 8 f3 on the device vs the oracle: 0..255 truncation and per-image error (integer work: bit-exact sums, the
    fp32 sqrt / division to 1e-6), and the bilinear resize + truncation (cv2.INTER_LINEAR's float64 rule, restated explicitly by the
    oracle and pinned by resize_cv2_rule.npz)."""
    from oracle import seg2eye_oracle as O
    from seg2eye_amd import ops, synthetic as syn
    from seg2eye_amd.networks.loss import MSECalculator, openEDSaccuracy
    from seg2eye_amd.postprocessor import ImageProcessor
    dev = _dev()
    a = torch.from_numpy(syn.make_batch(3, 96, 80, seed=501)['target']).to(dtype)
    b = torch.from_numpy(syn.make_batch(3, 96, 80, seed=502)['target']).to(dtype)
    ref = O.mse_for_tensors(a.float(), b.float())
    got = MSECalculator.calculate_mse_for_tensors(a.to(dev), b.to(dev)).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6)
    if dtype == torch.float32:                                   # and against the REAL reference's numbers
        np.testing.assert_allclose(got.numpy(), load_golden('openeds_metric')['mse_tensors'], rtol=1e-6)
    # resize to 640 x 400 + truncation: the device follows cv2's float64 rule operation by operation (csrc/metric.hip), so it
    # must give the oracle's integers EXACTLY wherever the float64 value is not within 1e-6 of an integer (there a last-bit
    # difference could truncate either way; none expected either)
    pre = O.to_255_pre_truncation(a.float())
    r_ref = pre.int()
    r_got = ImageProcessor.to_255resized_imagebatch(a.to(dev)).cpu()
    assert r_got.shape == (3, 1, 640, 400) and r_got.dtype == torch.uint8
    safe = (pre - pre.round()).abs() > 1e-6
    assert torch.equal(r_got.int()[safe], r_ref[safe]), int((r_got.int() != r_ref)[safe].sum())
    assert int((r_got.int() - r_ref).abs().max()) <= 1
    # images already in 0..255
    ia, ib = r_ref.to(torch.uint8), O.to_255_resized(b.float()).to(torch.uint8)
    np.testing.assert_allclose(MSECalculator.calculate_mse_for_images(ia.to(dev), ib.to(dev)).cpu().numpy(),
                               O.mse_for_images(ia, ib).numpy(), rtol=1e-6)
    np.testing.assert_allclose(float(openEDSaccuracy(ia[0].to(dev), ib[0].to(dev))), float(O.openeds_accuracy(ia[0], ib[0])), rtol=1e-6)
    assert ImageProcessor.to_255imagebatch(a.float()).equal(O.to_255(a.float()))



@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [(64, 96), (640, 384), (300, 200)])
def test_bilinear_resize_matches_torch(hw, dtype):
    """s2e_bilinear_resize_fwd / _bwd (the encoder's front end, encoder.py:54-55) against F.interpolate(mode='bilinear',
    align_corners=False) and its autograd, up- and down-scaling."""
    from seg2eye_amd import ops
    H, W = hw
    x = _rnd((3, 1, H, W), 51, torch.float32)
    gy = _rnd((3, 1, 256, 256), 52, dtype)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, size=(256, 256), mode='bilinear', align_corners=False)
    yr.backward(gy.double())
    xg = x.to(_dev()).requires_grad_(True)
    y = ops.bilinear_resize(xg, 256, 256, dtype)
    assert tuple(y.shape) == (3, 256, 256, 1) and y.dtype == dtype
    _close(y.permute(0, 3, 1, 2), yr, dtype, what='bilinear fwd')
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(_dev()))
    _close(xg.grad, xr.grad, torch.float32 if dtype == torch.float32 else dtype, what='bilinear bwd')


@pytest.mark.parametrize('dtype', DTYPES)
def test_spade_prepass_batches_label_convs_and_tables(dtype):
    """ops.SpadePrepass: from the third forward of a shape on, the label convs (mlp_shared + ReLU) and the per-class tables of
    every planned layer come out of ONE launch each (s2e_label_conv3x3_batch, s2e_spade_class_table_batch) -- the same bits as
    the per-layer launches -- and the plan is dropped when a tensor it points at moves."""
    from seg2eye_amd import ops
    from seg2eye_amd.synthetic import ellipse_labels
    dev = _dev()
    N, H, W = 2, 64, 64
    lab = torch.from_numpy(ellipse_labels(N, H, W, 5)[:, 0]).to(dev)
    layers = []
    for i, (h, w, c) in enumerate([(64, 64, 64), (32, 32, 64), (32, 32, 128), (16, 16, 192), (8, 8, 64)]):
        w_sh = _rnd((128, 4, 3, 3), 60 + i, torch.float32, 0.3).to(dev)
        b_sh = _rnd((128,), 70 + i, torch.float32, 0.1).to(dev)
        w_gb = _rnd((2 * c, 128, 3, 3), 80 + i, torch.float32, 0.03).to(dev)
        b_gb = _rnd((2 * c,), 90 + i, torch.float32, 0.1).to(dev)
        wp = ops.pack_weight(w_gb, dtype, 128, False, None)              # persistent for the test: held in `layers`
        layers.append((w_sh, b_sh, wp, b_gb, h, w, c))
    pre = ops.SpadePrepass()

    def forward():
        with pre.scope(lab, dtype):
            acts = [ops.SpadePrepass.actv(lab, w_sh, b_sh, N, H, W, h, w, 128, dtype) for w_sh, b_sh, wp, b_gb, h, w, c in layers]
            tabs = [ops.SpadePrepass.table(dtype, w_sh, b_sh, wp, b_gb, 4, 128, c, True) for w_sh, b_sh, wp, b_gb, h, w, c in layers]
        return acts, tabs
    a1, t1 = forward()                                                  # noted
    a2, t2 = forward()                                                  # learned
    plan = list(pre.plans.values())[0]
    assert plan is not None and plan['conv']['nb'] > len(layers) and len(plan['table']['entries']) == len(layers)
    a3, t3 = forward()                                                  # batched
    base = a3[0].untyped_storage().data_ptr()
    assert all(a.untyped_storage().data_ptr() == base for a in a3) and a1[0].untyped_storage().data_ptr() != a1[1].untyped_storage().data_ptr()
    for x, y, z in zip(a1, a2, a3):
        assert torch.equal(x, y) and torch.equal(x, z)
    for x, y, z in zip(t1, t2, t3):
        assert torch.equal(x, y) and torch.equal(x, z)
    # a weight that moved: the plan is dropped, the forward still gives the right numbers, and it is learned again
    layers[1][0].data = layers[1][0].data.clone()                       # (what re-homing a Parameter into a new arena does)
    a4, t4 = forward()
    assert list(pre.plans.values())[0] is None and all(torch.equal(x, y) for x, y in zip(a1, a4))
    forward()
    a6, _ = forward()
    assert all(torch.equal(x, y) for x, y in zip(a1, a6)) and a6[0].untyped_storage().data_ptr() == a6[1].untyped_storage().data_ptr()


def test_wgrad_c8_batch_matches_per_layer_launches():
    """s2e_wgrad_c8_batch (all queued mlp_shared weight / bias gradients of a step in one launch per slab shape, written
    straight in OIHW) against the per-layer path (s2e_conv2d_wgrad -> packed dW -> unpack), through ops.GradSink inside a
    ZeroPool scope as the trainer uses it; maps the batch does not take (8x8) fall back to the per-layer path."""
    from seg2eye_amd import ops
    dev = _dev()
    N, ncls = 3, 4
    pool = ops.ZeroPool(dev)
    cases = [(64, 64), (32, 32), (16, 16), (48, 80), (8, 8)]
    ins, ref = [], []
    for i, (h, w) in enumerate(cases):
        lab = _labels(N, h, w, 100 + i).to(dev)
        oh = ops.onehot_nhwc_raw(lab, None, h, w, ncls, 8, torch.bfloat16)
        dactv = nhwc(_rnd((N, 128, h, w), 110 + i, torch.bfloat16)).to(dev)
        dwp, db = ops.conv2d_wgrad_raw(oh, dactv, 3, 3, 1, 1, ops.ACT_NONE, True)
        ref.append((ops._unpack_dw(dwp, 128, ncls, 3, 3, 8).contiguous().clone(), db.clone()))
        ins.append((oh, dactv))
    dws = [torch.zeros(128, ncls, 3, 3, device=dev) for _ in cases]
    dbs = [torch.zeros(128, device=dev) for _ in cases]
    queued = []
    with pool.scope('t'):
        for (oh, dactv), dw, db in zip(ins, dws, dbs):
            queued.append(ops.GradSink.push_c8(oh, dactv, dw, db, ncls))
        assert len(pool.sink.c8) == sum(queued)
    assert queued == [True, True, True, True, False]           # 8x8: no 128-pixel slab covers it to 80 %
    torch.cuda.synchronize()
    for q, (rw, rb), dw, db in zip(queued, ref, dws, dbs):
        if q:
            assert float((dw - rw).abs().max()) <= 1e-4 * float(rw.abs().max()) + 1e-5
            assert float((db - rb).abs().max()) <= 1e-4 * float(rb.abs().max()) + 1e-5


@pytest.mark.parametrize('wgs', ['', '5', '37'])
def test_wgrad_batch_kernel_matches_fp64(wgs):
    """s2e_wgrad_batch (csrc/conv_wgrad_batch.hip): 36 jobs -- single-owner tiles, tiles shared between workgroups, ragged Cout,
    label-sparse rectangle lists (empty / odd / full), accumulation into non-zero dW, two launches -- against fp64 sums of the same
    bf16 operands (tools/check_wgrad_batch.py).  One workgroup per CU (the default: nearly every tile of these small cases is
    shared), 5 workgroups (long ranges: whole tiles and many segments per workgroup) and 37 (a mix)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if wgs:
        env['S2E_WGRAD_BATCH_WGS'] = wgs
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_wgrad_batch.py')], env=env, capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert 'worst relative error' in out.stdout


def test_wgrad_batch_through_the_sink():
    """Inside a trainer-step scope ops.conv2d_wgrad_raw QUEUES the patch-resident 3x3 weight gradients (GradSink.push_wgrad) and the
    scope's flush runs them as one launch; outside a scope -- and for shapes the batch does not take, or a dW queued twice -- the
    per-layer launch runs at once.  Both must give the same gradients."""
    from seg2eye_amd import ops
    dev = _dev()
    shapes = [(2, 32, 32, 64, 128), (2, 16, 16, 128, 256), (1, 64, 64, 64, 64), (2, 12, 20, 64, 64)]    # (the last: not made of 8 x 16 slabs)
    ins = []
    for i, (n, h, w, cin, cout) in enumerate(shapes):
        ins.append((nhwc(_rnd((n, cin, h, w), 300 + i, torch.bfloat16)).to(dev), nhwc(_rnd((n, cout, h, w), 320 + i, torch.bfloat16)).to(dev)))
    ref = []
    for x, gy in ins:                                           # no scope: launched one by one
        dw = torch.zeros(gy.shape[-1], 9 * x.shape[-1], device=dev)
        db = torch.zeros(gy.shape[-1], device=dev)
        ops.conv2d_wgrad_raw(x, gy, 3, 3, 1, 1, ops.ACT_NONE, True, db, dw_out=dw)
        ref.append((dw, db))
    pool = ops.ZeroPool(dev)
    outs = [(torch.zeros_like(dw), torch.zeros_like(db)) for dw, db in ref]
    with pool.scope('t'):
        for (x, gy), (dw, db) in zip(ins, outs):
            ops.conv2d_wgrad_raw(x, gy, 3, 3, 1, 1, ops.ACT_NONE, True, db, dw_out=dw)
        assert len(pool.sink.wg) == 3                            # the 12 x 20 map ran at once
        assert float(outs[0][0].abs().max()) == 0.0 and float(outs[3][0].abs().max()) > 0.0
        ops.conv2d_wgrad_raw(ins[0][0], ins[0][1], 3, 3, 1, 1, ops.ACT_NONE, True, outs[0][1], dw_out=outs[0][0])   # same dW again: at once
        assert len(pool.sink.wg) == 3 and float(outs[0][0].abs().max()) > 0.0
    torch.cuda.synchronize()
    for i, ((dw, db), (rw, rb)) in enumerate(zip(outs, ref)):
        k = 2.0 if i == 0 else 1.0
        assert float((dw - k * rw).abs().max()) <= 2e-5 * float(rw.abs().max()) * k + 1e-6, i
        assert float((db - k * rb).abs().max()) <= 2e-5 * float(rb.abs().max()) * k + 1e-6, i


# (n, hi, wi, cin, cout, k, stride, pad, bias, residual, lrelu): every mode of conv_plane.hip at >= 128 work items, ragged maps included
_PLANE_CFGS = [(4, 64, 64, 128, 256, 1, 1, 0, False, True, False), (5, 48, 80, 256, 64, 1, 1, 0, True, False, True),
               (8, 128, 128, 64, 128, 3, 2, 1, False, False, False), (8, 72, 104, 128, 64, 3, 2, 1, True, False, True),
               (4, 64, 64, 256, 512, 3, 2, 1, False, False, False),
               (8, 129, 129, 64, 128, 4, 2, 2, False, False, False), (16, 65, 65, 128, 256, 4, 2, 2, True, False, False),
               (16, 66, 50, 64, 64, 4, 2, 2, False, False, False)]


@pytest.mark.parametrize('cfg', _PLANE_CFGS)
def test_plane_conv_forward_and_gradients_match_fp64(cfg):
    """Round 6: netE's 3x3 stride-2 convs (reference models/networks/encoder.py:23-39), the PatchGAN's 4x4
    stride-2 convs (discriminator.py:84-96) and the learned 1x1 shortcuts (architecture.py:26-27,53-56) run in csrc/conv_plane.hip --
    parity-plane patch in LDS, weights in the PLANE pack layout straight into registers -- forward AND data gradient, through the
    same ops.conv2d the networks call (autograd included: the weight gradient goes through the multi-job / per-layer generic kernel).
    Against torch fp64 on the bf16-rounded operands."""
    from seg2eye_amd import ops
    from seg2eye_amd.ops import conv as oc
    n, hi, wi, cin, cout, k, st, pd, bias, res, lrelu = cfg
    dev, dt = _dev(), torch.bfloat16
    ho, wo = (hi + 2 * pd - k) // st + 1, (wi + 2 * pd - k) // st + 1
    act = ops.ACT_LRELU if lrelu else ops.ACT_NONE
    assert oc.plane_mode(dt, n, hi, wi, cin, ho, wo, cout, k, k, st, pd, False, ops.ACT_NONE, act, ops.AUX_NONE, res) > 0, 'forward not a plane shape'
    assert oc.plane_mode(dt, n, ho, wo, cout, hi, wi, cin, k, k, st, pd, True) > 0, 'data gradient not a plane shape'
    x = nhwc(_rnd((n, cin, hi, wi), 700, dt)).to(dev).requires_grad_(True)
    w = (_rnd((cout, cin, k, k), 701, torch.float32) / (cin * k * k) ** 0.5).to(dev).requires_grad_(True)
    b = _rnd((cout,), 702, torch.float32).to(dev).requires_grad_(True) if bias else None
    r = nhwc(_rnd((n, cout, ho, wo), 703, dt)).to(dev) if res else None
    proj = nhwc(_rnd((n, cout, ho, wo), 704, dt)).to(dev)
    y = ops.conv2d(x, w, b, r, st, pd, ops.ACT_NONE, act)
    (y.float() * proj.float()).sum().backward()
    xr = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(dt).double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, st, pd)
    if res:
        yr = yr + r.double().permute(0, 3, 1, 2)
    if lrelu:
        yr = F.leaky_relu(yr, 0.2)
    (yr * proj.double().permute(0, 3, 1, 2)).sum().backward()
    _close(y.permute(0, 3, 1, 2), yr, dt, what='plane forward')
    _close_kink(x.grad.permute(0, 3, 1, 2), xr.grad, dt, what='plane data gradient') if lrelu else _close(x.grad.permute(0, 3, 1, 2), xr.grad, dt, what='plane data gradient')
    _close_kink(w.grad, wr.grad, dt, what='weight gradient') if lrelu else _close(w.grad, wr.grad, dt, what='weight gradient')
    if bias:
        _close(b.grad, br.grad, dt, what='bias gradient')


@pytest.mark.parametrize('shape', [(128, 64, 3, 3), (192, 256, 1, 1), (64, 128, 4, 4), (200, 96, 3, 3)])
def test_plane_pack_layout(shape):
    """The PLANE weight layout (csrc/conv_plane.h): per (64 rows, 32-element K chunk, tap) a 4-KB block of four 16x16x32 fragments in
    register order with the row permutation that makes a lane's accumulators 8 consecutive channels.  Checked element by element
    against the definition, forward and transposed, from torch-order and from channels-last masters (the batched pack of a PackPlan
    uses the same device function)."""
    from seg2eye_amd import ops
    from seg2eye_amd.ops import conv as oc
    dev = _dev()
    cout, cin, kh, kw = shape
    taps = kh * kw
    w = _rnd(shape, 710, torch.float32).to(dev)
    wcl = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    sg = torch.tensor([1.7], device=dev)
    for tr in (False, True):
        rows, kd = (cin, cout) if tr else (cout, cin)
        if kd % 32:
            continue
        a = oc.pack_weight(w, torch.bfloat16, None, tr, sg, plane=True)
        b = oc.pack_weight(wcl, torch.bfloat16, None, tr, sg, plane=True)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), 'the two source layouts must pack to the same bytes'
        # definition: unit = ((g * nch + c) * taps + t) * 256 + f * 64 + L holds k = 32 c + 8 (L >> 4) .. + 7 of row
        # 64 g + 32 (f >> 1) + 8 ((L & 15) >> 2) + 4 (f & 1) + (L & 3)
        g64, nch = (rows + 63) // 64, kd // 32
        flat = a.reshape(-1).float().cpu().view(g64, nch, taps, 4, 64, 8)
        wd = (w / sg).to(torch.bfloat16).float().cpu()
        L = torch.arange(64)
        for f in range(4):
            row_in = 32 * (f >> 1) + 8 * ((L & 15) >> 2) + 4 * (f & 1) + (L & 3)                  # (64,)
            for g in range(g64):
                row = 64 * g + row_in
                ok = row < rows
                for c in range(nch):
                    k = 32 * c + 8 * (L >> 4)[:, None] + torch.arange(8)[None, :]             # (64, 8)
                    rr = row.clamp(max=rows - 1)[:, None].expand(64, 8)
                    ref = (wd[k, rr] if tr else wd[rr, k]).reshape(64, 8, taps).permute(2, 0, 1)   # (taps, 64, 8)
                    ref = ref * ok[None, :, None]
                    got = flat[g, c, :, f]
                    assert torch.equal(got, ref), (tr, f, g, c)


def test_c8_batch_rect_list_form():
    """Round 6: mlp_shared's weight gradient of a label-sparse SPADE backward (reference models/networks/normalization.py:85-89
    differentiated) walks the backward's work rectangles only: the rect-list form of s2e_wgrad_c8_batch must give the dense form's
    sums when gy is zero outside the listed rectangles -- and must not READ gy there (NaNs outside the list change nothing)."""
    import ctypes as C
    from seg2eye_amd import _lib as L
    dev = _dev()
    n, h, w, ncls = 4, 64, 64, 4
    g = torch.Generator().manual_seed(720)
    lab = torch.randint(0, ncls, (n, h, w), generator=g)
    oh = torch.zeros(n, h, w, 8)
    oh.scatter_(3, lab[..., None], 1.0)
    oh = oh.to(dev).to(torch.bfloat16)
    rects_all = n * (h // 16) * (w // 16)
    pick = torch.randperm(rects_all, generator=g)[:rects_all // 3].sort().values.to(torch.int32)
    mask = torch.zeros(rects_all, dtype=torch.bool)
    mask[pick.long()] = True
    pm = mask.view(n, h // 16, w // 16).repeat_interleave(16, 1).repeat_interleave(16, 2)
    gy = torch.randn(n, h, w, 128, generator=g) * pm[..., None]
    gy_d = gy.to(dev).to(torch.bfloat16)
    gy_nan = gy_d.clone()
    gy_nan[~pm.to(dev)] = float('nan')
    lst, cnt = pick.to(dev), torch.tensor([pick.numel(), 0], dtype=torch.int32, device=dev)
    outs = []
    for gyt, use_list in ((gy_d, False), (gy_d, True), (gy_nan, True)):
        dw = torch.zeros(128, ncls, 3, 3, device=dev)
        db = torch.zeros(128, device=dev)
        job = (L.WgradC8Job * 1)()
        job[0].x, job[0].gy, job[0].dw_oihw, job[0].dbias = oh.data_ptr(), gyt.data_ptr(), dw.data_ptr(), db.data_ptr()
        job[0].H, job[0].W, job[0].ncls = h, w, ncls
        if use_list:
            job[0].rect_list, job[0].rect_count = lst.data_ptr(), cnt.data_ptr()
        wsb = L.lib().s2e_wgrad_c8_batch_workspace_bytes(n, C.byref(job), 1)
        ws = torch.empty(wsb // 4, device=dev)
        L.check(L.lib().s2e_wgrad_c8_batch(L.S2E_BF16, n, C.byref(job), 1, ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream), 's2e_wgrad_c8_batch')
        outs.append((dw, db))
    torch.cuda.synchronize()
    ref = F.conv2d(oh[..., :ncls].double().permute(3, 0, 1, 2), gy_d.double().permute(3, 0, 1, 2), None, 1, 1).permute(1, 0, 2, 3)     # (128, ncls, 3, 3)
    for dw, db in outs:
        assert torch.isfinite(dw).all() and torch.isfinite(db).all()
        assert float((dw.double() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-5
        assert float((db.double() - gy_d.double().sum((0, 1, 2))).abs().max()) <= 2e-4 * float(gy_d.double().sum((0, 1, 2)).abs().max()) + 1e-4


@pytest.mark.parametrize('dtype', DTYPES)
def test_shard_sum_matches_torch(dtype):
    """The owner sum of the 'direct' gradient exchange (seg2eye_amd/distributed.py): fp32 accumulation in rank order, one rounding."""
    from seg2eye_amd import _lib as L
    dev = _dev()
    world, shard = 8, 4096 + 8
    recv = _rnd((world, shard), 730, dtype).to(dev)
    out = torch.empty(shard, dtype=dtype, device=dev)
    L.check(L.lib().s2e_shard_sum(L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32, recv.data_ptr(), out.data_ptr(), world, shard,
                                  torch.cuda.current_stream().cuda_stream), 's2e_shard_sum')
    acc = torch.zeros(shard, dtype=torch.float32, device=dev)
    for r in range(world):
        acc += recv[r].float()
    assert torch.equal(out, acc.to(dtype))


def test_generic_wgrad_multi_matches_per_layer_launches():
    """Round 6: inside a trainer-step scope the GENERIC weight gradients (1x1, stride 2, 4x4, small maps) are queued too
    (GradSink.push_gwg) and the flush runs them as one multi-job launch + one reduction launch (s2e_conv2d_wgrad_multi); the sums
    must be those of the per-layer launches (s2e_conv2d_wgrad), job by job -- split jobs with partial tiles, unsplit ones, with and
    without a bias gradient, a dW used twice (the second use runs at once)."""
    from seg2eye_amd import ops
    dev = _dev()
    #        n   hi  wi  cin  cout k  s  p  bias
    cfgs = [(4, 64, 64, 64, 128, 3, 2, 1, False), (2, 32, 32, 256, 128, 1, 1, 0, False), (2, 33, 33, 64, 128, 4, 2, 2, True),
            (8, 8, 8, 128, 256, 3, 1, 1, True), (4, 17, 17, 128, 64, 4, 1, 2, False), (16, 128, 128, 64, 128, 3, 2, 1, False)]
    ins = []
    for i, (n, hi, wi, cin, cout, k, st, pd, bias) in enumerate(cfgs):
        ho, wo = (hi + 2 * pd - k) // st + 1, (wi + 2 * pd - k) // st + 1
        ins.append((nhwc(_rnd((n, cin, hi, wi), 500 + i, torch.bfloat16)).to(dev), nhwc(_rnd((n, cout, ho, wo), 520 + i, torch.bfloat16)).to(dev)))

    def run(deferred):
        outs = []
        for (n, hi, wi, cin, cout, k, st, pd, bias), (x, gy) in zip(cfgs, ins):
            dw = torch.zeros(cout, k * k * cin, device=dev)
            db = torch.zeros(cout, device=dev) if bias else None
            ops.conv2d_wgrad_raw(x, gy, k, k, st, pd, ops.ACT_NONE, bias, db, dw_out=dw, defer_ok=deferred)
            outs.append((dw, db))
        return outs
    ref = run(False)
    pool = ops.ZeroPool(dev)
    with pool.scope('t'):
        outs = run(True)
        assert len(pool.sink.gwg) == len(cfgs) and float(outs[0][0].abs().max()) == 0.0          # queued, nothing written yet
        x, gy = ins[1]
        ops.conv2d_wgrad_raw(x, gy, 1, 1, 1, 0, ops.ACT_NONE, False, None, dw_out=outs[1][0], defer_ok=True)   # same dW again: at once
        assert len(pool.sink.gwg) == len(cfgs) and float(outs[1][0].abs().max()) > 0.0
    torch.cuda.synchronize()
    for i, ((dw, db), (rw, rb)) in enumerate(zip(outs, ref)):
        kf = 2.0 if i == 1 else 1.0
        assert float((dw - kf * rw).abs().max()) <= 3e-5 * float(rw.abs().max()) * kf + 1e-6, (i, float((dw - kf * rw).abs().max()), float(rw.abs().max()))
        if rb is not None:
            assert float((db - rb).abs().max()) <= 3e-5 * float(rb.abs().max()) + 1e-6, i


def test_c8_first_layer_kernels_against_fp64():
    """conv_c8.hip: the PatchGAN's first layer (8 -> 64 channels, 4x4 stride 2 pad 2) forward (bias / LeakyReLU / residual), data gradient and
    weight + bias gradient on even, odd and ragged maps against torch fp64 on the bf16-rounded operands, through s2e_conv2d and
    s2e_conv2d_wgrad_multi (tools/check_c8.py asserts; the padding channels of the data gradient must be exactly zero)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop('S2E_CONV_C8', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_c8.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'parity ok' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count('kind 6') == 5, r.stdout


def test_flat_wgrad_kinds_against_fp64():
    """conv_wgrad_flat.hip, every kind (1x1, 3x3 stride 1 / 2, 4x4 stride 1 / 2) on small, ragged and multi-tile shapes, with bias
    gradients: dW and db through s2e_conv2d_wgrad_multi against torch fp64 on the bf16-rounded operands (tools/check_wgrad_flat.py; the
    kinds that are off by default are switched on through S2E_WGRAD_FLAT, which the library reads once -- hence the child process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, S2E_WGRAD_FLAT='62')
    env.pop('S2E_WF_NOEPI', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_wgrad_flat.py')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'parity ok' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    kinds = [int(l.split('kind')[1].split()[0]) for l in r.stdout.splitlines() if ' kind ' in l]
    assert sorted(set(kinds)) == [1, 2, 3, 4, 5], kinds


@pytest.mark.parametrize('order', ['queued_then_immediate', 'immediate_then_queued'])
def test_wgrad_sink_fresh_arena_view_used_at_two_shapes(order):
    """a dW that is a view of a gradient arena zero_grad has just cleared (`grad_is_fresh`: single-owner tiles of the batched
    launch are STORED, not added) and whose weight is used twice in one scope, once at a shape the batch takes (queued) and once at
    one it declines (the per-layer kernel accumulates at once) -- in either order the flush must ADD to what is there."""
    from seg2eye_amd import ops
    dev = _dev()
    cin = cout = 64
    a = (nhwc(_rnd((2, cin, 32, 32), 401, torch.bfloat16)).to(dev), nhwc(_rnd((2, cout, 32, 32), 402, torch.bfloat16)).to(dev))     # batch shape
    b = (nhwc(_rnd((2, cin, 12, 20), 403, torch.bfloat16)).to(dev), nhwc(_rnd((2, cout, 12, 20), 404, torch.bfloat16)).to(dev))     # not 8 x 16 slabs
    ref = torch.zeros(cout, 9 * cin, device=dev)
    for x, gy in (a, b):                                        # no scope: both accumulate at once
        ops.conv2d_wgrad_raw(x, gy, 3, 3, 1, 1, ops.ACT_NONE, False, None, dw_out=ref)
    arena = torch.zeros(cout * 9 * cin + 64, device=dev)
    ops.ZeroPool.arena_zeroed(arena)                            # (what optim.FlatAdam.zero_grad reports)
    dw = arena[:cout * 9 * cin].view(cout, 9 * cin)
    pool = ops.ZeroPool(dev)
    with pool.scope('t'):
        assert ops.ZeroPool.grad_is_fresh(dw)
        for x, gy in ((a, b) if order == 'queued_then_immediate' else (b, a)):
            ops.conv2d_wgrad_raw(x, gy, 3, 3, 1, 1, ops.ACT_NONE, False, None, dw_out=dw)
        assert len(pool.sink.wg) == 1 and pool.sink.wg[0][6] is False        # queued, and told to add
    torch.cuda.synchronize()
    assert float((dw - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize('duo', ['512', '0'])
def test_patch_conv_kernels_match_torch_at_bench_shapes(duo):
    """The two patch-resident 3x3 kernels at the bench's shapes against torch's own convolution in fp32 (tools/check_duo.py):
    forward with bias + residual + LeakyReLU, data-gradient with the ReLU mask, and the fused [gamma | beta] conv + SPADE+Style
    modulation, dense and through an odd-length rectangle list.  S2E_CONV_DUO=512 (the default): csrc/conv_duo.hip, two
    workgroups per CU, takes them; =0: csrc/conv_patch.hip runs the same checks (it keeps every shape the duo plan declines)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, S2E_CONV_DUO=duo)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_duo.py')], env=env, capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert 'worst relative error' in out.stdout


@pytest.mark.parametrize('cfg', [(16, 129, 129, 64, 128, False), (16, 65, 65, 128, 256, True), (8, 129, 129, 64, 128, True), (16, 130, 134, 256, 64, False),
                                 (12, 130, 127, 64, 40, True)])
def test_stride2_patch_conv_matches_torch(cfg):
    """The PatchGAN's 4x4 stride-2 pad-2 layers (discriminator.py:84-96) through the space-to-depth view (csrc/conv_patch.hip, S2D):
    forward with bias + LeakyReLU and the data gradient with the LeakyReLU mask, against torch's convolution on the same bf16
    operands in fp32, on odd and even maps, 64- and 128-column tiles; the plan must route these shapes to the patch kernel."""
    import torch.nn.functional as F
    from seg2eye_amd import ops
    from seg2eye_amd import _lib as L
    import ctypes as C
    n, H, W, cin, cout, masked = cfg
    dev, dt = _dev(), torch.bfloat16
    torch.manual_seed(11)
    ho, wo = H // 2 + 1, W // 2 + 1
    x = torch.randn(n, H, W, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 4, 4, device=dev) / (cin * 16) ** 0.5
    b = torch.randn(cout, device=dev)
    gy = torch.randn(n, ho, wo, cout, device=dev).to(dt)
    d = L.ConvDesc(n, H, W, cin, ho, wo, cout, 4, 4, 2, 2, 0, ops.ACT_NONE, ops.ACT_LRELU, ops.AUX_NONE)
    assert L.lib().s2e_conv2d_kernel_kind(L.S2E_BF16, C.byref(d)) == 2                  # S2E_KERNEL_PATCH
    dt_ = L.ConvDesc(n, ho, wo, cout, H, W, cin, 4, 4, 2, 2, 1, ops.ACT_NONE, ops.ACT_NONE, ops.AUX_LRELU_GRAD if masked else ops.AUX_NONE)
    assert L.lib().s2e_conv2d_kernel_kind(L.S2E_BF16, C.byref(dt_)) == (2 if cin >= 64 and cout % 64 == 0 else 0)
    wb = w.to(dt).float()
    y = ops.conv2d_raw(x, ops.pack_weight(w, dt, cin, False), b, None, None, (ho, wo, cout), 4, 4, 2, 2, False, ops.ACT_NONE, ops.ACT_LRELU)
    ref = F.leaky_relu(F.conv2d(x.float().permute(0, 3, 1, 2), wb, b, stride=2, padding=2), 0.2).permute(0, 2, 3, 1)
    err = float((y.float() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 6e-3, err                                                            # (one bf16 rounding of the result)
    gx = ops.conv2d_raw(gy, ops.pack_weight(w, dt, cin, True), None, None, x if masked else None, (H, W, cin), 4, 4, 2, 2, True,
                        ops.ACT_NONE, ops.ACT_NONE, ops.AUX_LRELU_GRAD if masked else ops.AUX_NONE)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.conv2d(xr, wb, None, stride=2, padding=2).backward(gy.float().permute(0, 3, 1, 2))
    gref = xr.grad.permute(0, 2, 3, 1)
    assert bool(torch.isfinite(gx.float()).all()) and bool(torch.isfinite(gref).all())
    if masked:
        gref = gref * torch.where(x.float() > 0, 1.0, 0.2)
    gerr = float((gx.float() - gref).abs().max()) / float(gref.abs().max())
    assert gerr <= 6e-3, gerr


@pytest.mark.parametrize('cfg', [(4, 128, 128, 256, 128, True), (2, 256, 256, 128, 64, False), (8, 64, 64, 512, 256, True), (3, 48, 48, 64, 128, False)])
def test_conv_epilogue_instance_norm_statistics(cfg):
    """SURVEY 7 step 5 (round 4): the InstanceNorm statistics of a conv's output from the conv kernel's own epilogue
    (s2e_conv2d_stats -> s2e_in_stats_from_partials) equal the statistics pass over the stored output (s2e_in_stats), with and
    without a residual, for 128- and 64-channel tiles; a shape whose kernel has no such epilogue leaves the holder empty."""
    from seg2eye_amd import ops
    from seg2eye_amd import _lib as L
    n, H, W, cin, cout, with_res = cfg
    dev, dt = _dev(), torch.bfloat16
    torch.manual_seed(5)
    x = torch.randn(n, H, W, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    b = torch.randn(cout, device=dev)
    res = (torch.randn(n, H, W, cout, device=dev) + 0.5).to(dt) if with_res else None
    wp = ops.pack_weight(w, dt, cin, False)
    holder = []
    y = ops.conv2d_raw(x, wp, b, res, None, (H, W, cout), 3, 3, 1, 1, stats_out=holder)
    y0 = ops.conv2d_raw(x, wp, b, res, None, (H, W, cout), 3, 3, 1, 1)
    assert torch.equal(y, y0)                                           # the statistics epilogue changes no output bit
    if H == 48:
        assert not holder                                               # (ragged rectangles: conv_patch.hip's kernel, no statistics)
        return
    assert len(holder) == 1 and tuple(holder[0].shape) == (n, cout, 2)
    ref = ops.in_stats(y)
    mean_scale = float(ref[..., 0].abs().max()) + float(1.0 / ref[..., 1].min())
    assert float((holder[0][..., 0] - ref[..., 0]).abs().max()) <= 2e-6 * mean_scale
    assert float(((holder[0][..., 1] - ref[..., 1]) / ref[..., 1]).abs().max()) <= 1e-5
    # and against fp64 torch on the stored values
    yf = y.double().view(n, H * W, cout)
    np.testing.assert_allclose(holder[0][..., 0].cpu().numpy(), yf.mean(1).cpu().numpy(), atol=2e-6 * mean_scale)
    np.testing.assert_allclose(holder[0][..., 1].cpu().numpy(), (yf.var(1, unbiased=False) + 1e-5).rsqrt().cpu().numpy(), rtol=1e-5)


def test_conv_stream_kernel_every_shape_matches_torch():
    """csrc/conv_stream.hip on EVERY shape it can run (S2E_CONV_STREAM=2; by default it takes the long-K tiles only): forward, the
    stride-2 data-gradients by parity class, split tiles + fix-up -- tools/bench_tail.py checks each against torch in fp32."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, S2E_CONV_STREAM='2', S2E_WGRAD_PARTIAL='2')
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_tail.py'), '--iters', '2'], env=env, capture_output=True,
                         text=True, timeout=900, cwd=root)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize('cfg', [(8, 16, 21888), (1, 16, 300), (4, 8, 1000), (3, 32, 513), (32, 64, 700)])
def test_style_fc_kernels_match_torch(cfg):
    """csrc/style_fc.hip vs the torch expression of the stacked ApplyStyle FCs (normalization.py:144-169), forward and backward;
    the backward's dw (a two-level reduction over S with a last-block fold) must be bit-identical launch to launch."""
    from seg2eye_amd import _lib as L
    N, K, S = cfg
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    w = torch.randn(N, K, generator=g).to(dev)
    W = (torch.randn(S, K, generator=g) * 0.3).to(dev)
    b = (torch.randn(S, generator=g) * 0.1).to(dev)
    dbig = torch.randn(N, S, generator=g).to(dev)
    gbig = torch.randn(N, S, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    assert L.lib().s2e_style_fc_supported(N, K)
    big = torch.empty(N, S, device=dev)
    L.check(L.lib().s2e_style_fc_fwd(w.data_ptr(), W.data_ptr(), b.data_ptr(), big.data_ptr(), N, K, S, 0.2, st), 'fwd')
    ref = F.leaky_relu(torch.addmm(b.double(), w.double(), W.double().t()), 0.2)
    assert torch.allclose(big.double(), ref, rtol=1e-5, atol=1e-5)
    wsb = L.lib().s2e_style_fc_bwd_workspace_bytes(N, K, S)
    outs = []
    for gb_in in (None, gbig):
        for rep in range(2):
            gW = torch.full((S, K), 0.5, device=dev)
            gb = torch.full((S,), -0.25, device=dev)
            dw = torch.empty(N, K, device=dev)
            ws = torch.full((wsb // 4,), float('nan'), device=dev)                      # (no initialisation needed)
            L.check(L.lib().s2e_style_fc_bwd(dbig.data_ptr(), None if gb_in is None else gb_in.data_ptr(), big.data_ptr(), w.data_ptr(),
                                             W.data_ptr(), gW.data_ptr(), gb.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, N, K, S, 0.2, st), 'bwd')
            outs.append((gW, gb, dw))
        d = dbig.double() if gb_in is None else (dbig + gb_in).double()
        dpre = torch.where(big.double() > 0, d, 0.2 * d)
        assert torch.allclose(outs[-1][0].double(), 0.5 + dpre.t() @ w.double(), rtol=1e-4, atol=1e-4)
        assert torch.allclose(outs[-1][1].double(), -0.25 + dpre.sum(0), rtol=1e-4, atol=1e-4)
        assert torch.allclose(outs[-1][2].double(), dpre @ W.double(), rtol=1e-4, atol=1e-3 * S ** 0.5)
        assert torch.equal(outs[-1][2], outs[-2][2])
    # dw not wanted: no workspace needed
    gW = torch.zeros(S, K, device=dev)
    gb = torch.zeros(S, device=dev)
    L.check(L.lib().s2e_style_fc_bwd(dbig.data_ptr(), None, big.data_ptr(), w.data_ptr(), W.data_ptr(), gW.data_ptr(), gb.data_ptr(),
                                     None, None, 0, N, K, S, 0.2, st), 'bwd')
    assert not L.lib().s2e_style_fc_supported(33, 16) and not L.lib().s2e_style_fc_supported(8, 12)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 64, 64), (8, 128, 128, 128), (3, 40, 50, 2064), (1, 256, 256, 8)])
def test_in_stats_one_launch_equals_two_launches_bitwise(shape, dtype):
    """The fold of the per-block partial sums done by the last row-walking block (counters given) gives the bits of the separate
    finalize launch, launch after launch, and leaves the counters zero."""
    from seg2eye_amd import _lib as L
    dev = _dev()
    n, h, w, c = shape
    x = (_rnd(shape, 3, dtype) * 1.7 + 0.3).to(dev)
    dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
    st = torch.cuda.current_stream().cuda_stream
    wsb = L.lib().s2e_in_stats_workspace_bytes(dt, n, h * w, c)
    ncnt = L.lib().s2e_in_stats_counters(dt, n, h * w, c)
    assert ncnt > 0
    res = []
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=dev)
    for counters in (None, cnt, cnt, cnt):
        ws = torch.full((wsb // 8,), float('nan'), dtype=torch.float64, device=dev)
        stats = torch.empty(n, c, 2, device=dev)
        L.check(L.lib().s2e_in_stats(dt, x.data_ptr(), n, h * w, c, 1e-5, ws.data_ptr(), stats.data_ptr(),
                                     None if counters is None else counters.data_ptr(), st), 'in_stats')
        res.append((stats, ws[:n * c * 2].clone()))
        assert int(cnt.abs().sum()) == 0
    for stats, sums in res[1:]:
        assert torch.equal(stats, res[0][0]) and torch.equal(sums, res[0][1])
    ref = x.double().cpu().view(n, h * w, c)
    assert torch.allclose(res[0][0][..., 0].double().cpu(), ref.mean(1), rtol=1e-4, atol=1e-4)


def test_discriminator_input_and_split_halves():
    """ops.d_input == the reference's cat([cat(one_hot, fake); cat(one_hot, real)]) with the gradient reaching `fake` only;
    ops.split_halves == two slices, including a half that receives no gradient."""
    from seg2eye_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    n, H, W = 3, 20, 28
    label = torch.randint(0, 4, (n, H, W), generator=g, dtype=torch.uint8).to(dev)
    fake = torch.randn(n, 1, H, W, generator=g).to(dev).requires_grad_(True)
    real = torch.randn(n, 1, H, W, generator=g).to(dev)
    x = ops.d_input(label, fake, real, 4, 8)
    oh = F.one_hot(label.long(), 4).float()
    ref = torch.zeros(2 * n, H, W, 8, device=dev)
    ref[:n, ..., :4], ref[n:, ..., :4] = oh, oh
    ref[:n, ..., 4], ref[n:, ..., 4] = fake.detach()[:, 0], real[:, 0]
    assert torch.equal(x, ref)
    gx = torch.randn(2 * n, H, W, 8, generator=g).to(dev)
    x.backward(gx)
    assert torch.equal(fake.grad, gx[:n, ..., 4].unsqueeze(1))
    t = torch.randn(6, 1, 5, 7, generator=g).to(dev).requires_grad_(True)
    a, b = ops.split_halves(t)
    assert torch.equal(a, t[:3]) and torch.equal(b, t[3:])
    (a * 2.0).sum().backward()
    assert torch.equal(t.grad[:3], torch.full_like(t[:3], 2.0)) and float(t.grad[3:].abs().sum()) == 0.0
    t.grad = None
    a, b = ops.split_halves(t)
    (a.sum() + 3.0 * b.sum()).backward()
    assert torch.equal(t.grad[3:], torch.full_like(t[3:], 3.0)) and torch.equal(t.grad[:3], torch.ones_like(t[:3]))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(256, 128, 3, 3), (48, 24, 3, 3), (130, 64, 4, 4), (72, 8, 1, 1), (1, 512, 4, 4), (32, 16, 5, 5)])
def test_pack_of_channels_last_master_equals_pack_of_torch_order(shape, dtype):
    """s2e_pack_conv_weight(transposed | 2) -- the streaming convert / the per-tap tile transpose from a [co][tap][ci] master --
    writes the very matrix the OIHW packers write (padding rows, K tail and the division by sigma included)."""
    from seg2eye_amd import ops
    dev = _dev()
    w = _rnd(shape, 31, torch.float32, 0.3).to(dev)
    wcl = w.contiguous(memory_format=torch.channels_last)
    sigma = torch.tensor([1.7], device=dev)
    for tr in (False, True):
        for sg in (None, sigma):
            a = ops.pack_weight(w, dtype, None, tr, sg)
            b = ops.pack_weight(wcl, dtype, None, tr, sg)
            assert a.shape == b.shape and torch.equal(a, b), (shape, tr, sg is not None)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(32, 4, 512, 16), (3, 4, 64, 5), (64, 2, 24, 32), (1, 4, 8, 1)])
def test_fc_head_matches_torch_linear_on_leaky_relu_of_nchw_flatten(cfg, dtype):
    """ops.fc_head == fc(LeakyReLU(x).view(M, -1)) with torch's (c, y, x) flattening (encoder.py:68-71), forward, dx, dW, db;
    gradients accumulate into an existing .grad."""
    from seg2eye_amd import ops
    M, so, C, N = cfg
    dev = _dev()
    x = _rnd((M, C, so, so), 41, dtype)
    lin = torch.nn.Linear(C * so * so, N).double()
    xr = x.double().requires_grad_(True)
    yr = lin(F.leaky_relu(xr, 0.2).view(M, -1))
    gy = _rnd((M, N), 42, torch.float32)
    yr.backward(gy.double())
    w = torch.nn.Parameter(lin.weight.detach().float().to(dev))
    b = torch.nn.Parameter(lin.bias.detach().float().to(dev))
    w.grad, b.grad = torch.full_like(w, 0.25), torch.full_like(b, -0.5)
    xg = nhwc(x).to(dev).requires_grad_(True)
    y = ops.fc_head(xg, w, b, 0.2)
    assert y is not None and y.dtype == torch.float32
    _close(y, yr, torch.float32, what='fc head y')
    y.backward(gy.to(dev))
    _close(nchw(xg.grad), xr.grad, dtype, what='fc head dx')
    _close(w.grad - 0.25, lin.weight.grad, torch.float32, what='fc head dW')
    _close(b.grad + 0.5, lin.bias.grad, torch.float32, what='fc head db')
    assert ops.fc_head(torch.zeros(65, so, so, C, device=dev, dtype=dtype), w, b) is None
    # the one-block-per-sample form of the C ABI (no workspace) gives the same numbers as the two-launch form ops.fc_head uses
    from seg2eye_amd import _lib as L
    y1 = torch.empty(M, N, dtype=torch.float32, device=dev)
    xd = nhwc(x).to(dev)
    L.check(L.lib().s2e_fc_head_fwd(ops._dt(xd), xd.data_ptr(), w.data_ptr(), b.data_ptr(), y1.data_ptr(), M, so * so, C, N, 0.2, None, 0,
                                    torch.cuda.current_stream().cuda_stream), 's2e_fc_head_fwd')
    _close(y1, yr, torch.float32, what='fc head y (no workspace)')
