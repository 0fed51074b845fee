"""Convolutions: weight packs, the raw launchers (s2e_conv2d / s2e_conv2d_wgrad), Conv2dFn with its data / weight / bias gradients
(spectral norm's chain rule included), the live-prefix gate of the G step's discriminator pass, the encoder's FC head."""
import ctypes as C

import torch

from .. import _lib as L
from .. import packing
from .._lib import ConvDesc, ACT_NONE, ACT_LRELU, ACT_TANH, AUX_NONE, AUX_LRELU_GRAD
from .core import GradSink, IN_EPS, LaunchProfiler, ZeroPool, _cl_dense, _cl_rows, _dt, _grad_dst, _need, _p, _stream, colsum


# ------------------------------------------------------------------------------ raw launchers

PLANE_LAYOUT = 4            # S2E_PACK_PLANE: the weight layout of s2e_conv2d_plane (csrc/conv_plane.h)


def pack_weight(w_oihw, dtype, cin_pad=None, transposed=False, sigma=None, plane=False):
    """OIHW fp32 -> MFMA B-operand matrix in the compute dtype; divided by the device scalar `sigma`
    (spectral norm) on the fly when given.  plane: the PLANE layout (the weight operand of s2e_conv2d_plane: plane_mode())."""
    w = w_oihw.detach()
    cout, cin, kh, kw = w.shape
    cin_pad = cin if cin_pad is None else cin_pad
    if plane:
        if dtype != torch.bfloat16 or cin_pad != cin:
            raise ValueError('pack_weight: the plane layout is bf16 without channel padding')
        cl = w.dtype == torch.float32 and not w.is_contiguous() and _cl_dense(w)
        if not cl and (w.dtype != torch.float32 or not w.is_contiguous()):
            w = w.float().contiguous()
        _need(_cl_rows(w) if cl else w, sigma)
        rows = cin if transposed else cout
        out = torch.empty((rows + 63) // 64 * 64, kh * kw * (cout if transposed else cin), dtype=dtype, device=w.device)
        L.check(L.lib().s2e_pack_conv_weight(L.S2E_BF16, _p(w), _p(out), _p(sigma), cout, cin, kh, kw, cin, int(bool(transposed)) | (2 if cl else 0) | PLANE_LAYOUT,
                                             _stream()), 's2e_pack_conv_weight')
        return out
    # a weight stored channels-last (optim.FlatAdam) is packed from where it lies: rows in, rows out
    cl = w.dtype == torch.float32 and not w.is_contiguous() and _cl_dense(w) and cin_pad == cin and cin % 8 == 0
    if not cl and (w.dtype != torch.float32 or not w.is_contiguous()):
        w = w.float().contiguous()
    _need(_cl_rows(w) if cl else w, sigma)
    transposed = int(bool(transposed)) | (2 if cl else 0)
    dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
    lib = L.lib()
    rows = lib.s2e_conv_cout_pad(cin_pad if (transposed & 1) else cout)
    kpad = lib.s2e_conv_k_pad(dt, kh * kw * (cout if (transposed & 1) else cin_pad))
    out = torch.empty(rows, kpad, dtype=dtype, device=w.device)
    L.check(lib.s2e_pack_conv_weight(dt, _p(w), _p(out), _p(sigma), cout, cin, kh, kw, cin_pad, int(transposed), _stream()),
            's2e_pack_conv_weight')
    return out


def packed_weight(w, dtype, cin_pad, transposed, sigma, plan, generation=None, stable=True, plane=False):
    """The packed matrix from the network's PackPlan (packing.py) when it has one for THIS forward, else an
    individual pack -- which also teaches the plan, so the next forward packs it in the batched launch.
    stable=False: `w` is a temporary (its address means nothing next time): never recorded.
    plane: the PLANE layout (the conv runs in s2e_conv2d_plane: plane_mode())."""
    if plan is not None and stable:
        wp = plan.lookup(w, dtype, cin_pad, transposed, generation, plane)
        if wp is not None:
            return wp
        if generation is None or generation == plan.generation:
            plan.record(w, dtype, cin_pad, transposed, sigma, plane)
    return pack_weight(w, dtype, cin_pad, transposed, sigma, plane)


_PLANE_MODES = {}


def plane_mode(x_dtype, n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, transposed, in_act=ACT_NONE, out_act=ACT_NONE, aux_mode=AUX_NONE,
               has_res=False):
    """0, or the mode (> 0) in which the plane-patch kernel (s2e_conv2d_plane) runs this launch: its weight operand is then packed
    in the PLANE layout (pack_weight(..., plane=True)) and conv2d_raw is called with plane=True.  Memoised per shape."""
    if x_dtype != torch.bfloat16 or (has_res and aux_mode != AUX_NONE):
        return 0
    key = (n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, int(transposed), in_act, out_act, aux_mode)
    v = _PLANE_MODES.get(key)
    if v is None:
        d = ConvDesc(*key)
        v = _PLANE_MODES[key] = int(L.lib().s2e_conv2d_plane_supported(L.S2E_BF16, C.byref(d)))
    return v


# profiler families = the kernel s2e_conv2d / s2e_conv2d_wgrad choose for the shape (S2E_KERNEL_GENERIC / SMALL / PATCH)
_CONV_FAMILY = ('conv_igemm', 'conv_small', 'conv_patch')
_WGRAD_FAMILY = ('conv_wgrad', 'conv_wgrad_small', 'conv_wgrad_patch')


_CONV_PLANS = {}


def _conv_plan(wgrad, dt, *shape):
    """(s2e_conv_desc, workspace bytes) of a launch, memoised per shape: a step repeats the same ~150 shapes, and building the
    ctypes structure + asking the library for the workspace size cost ~3 us of host time per launch."""
    key = (wgrad, dt) + shape
    ent = _CONV_PLANS.get(key)
    if ent is None:
        d = ConvDesc(*shape)
        wsb = (L.lib().s2e_conv2d_wgrad_workspace_bytes if wgrad else L.lib().s2e_conv2d_workspace_bytes)(dt, C.byref(d))
        if len(_CONV_PLANS) > 8192:
            _CONV_PLANS.clear()
        ent = _CONV_PLANS[key] = (d, wsb)
    return ent
_CONV_STATS_SLOTS = {}


def _conv_stats_slots(dt, d, *shape):
    """s2e_conv2d_stats_slots, memoised per shape."""
    key = (dt,) + shape
    v = _CONV_STATS_SLOTS.get(key)
    if v is None:
        v = _CONV_STATS_SLOTS[key] = int(L.lib().s2e_conv2d_stats_slots(dt, C.byref(d)))
    return v


def conv2d_raw(x, wp, bias, residual, aux, out_hw_c, kh, kw, stride, pad, transposed=False,
               in_act=ACT_NONE, out_act=ACT_NONE, aux_mode=AUX_NONE, out=None, stats_out=None, plane=False):
    """plane: wp is in the PLANE layout and the launch goes to s2e_conv2d_plane (the caller asked plane_mode()).
    stats_out: None, or a list that receives the InstanceNorm statistics (N, Cout, 2) {mean, rstd} of the result when the
    kernel this shape takes produces their partial sums in its epilogue (s2e_conv2d_stats; then the caller needs no pass over y);
    left empty otherwise."""
    _need(x, wp, bias, residual, aux)
    n, hi, wi, cin = x.shape
    ho, wo, cout = out_hw_c
    y = torch.empty(n, ho, wo, cout, dtype=x.dtype, device=x.device) if out is None else out
    if out is not None and (tuple(out.shape) != (n, ho, wo, cout) or out.dtype != x.dtype or not out.is_contiguous()):
        raise ValueError('conv2d_raw: out must be a contiguous %s tensor with the shape of the result' % (x.dtype,))
    dt = _dt(x)
    d, wsb = _conv_plan(False, dt, n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, int(transposed), in_act, out_act, aux_mode)
    # algorithmic FLOPs = those of the forward conv this launch computes or differentiates (a stride-2
    # data-gradient executes 4x that on structural zeros; not counted)
    pix = hi * wi if transposed else ho * wo
    flops = 2.0 * n * pix * cin * cout * kh * kw
    if plane:
        LaunchProfiler.run('conv_plane', flops, lambda: L.check(
            L.lib().s2e_conv2d_plane(dt, _p(x), _p(wp), _p(bias), _p(residual), _p(aux), _p(y), C.byref(d), _stream()), 's2e_conv2d_plane'),
            tag=lambda: '%s n%d %dx%d c%d->%d k%d s%d' % ('D' if transposed else 'F', n, hi, wi, cin, cout, kh, stride),
            nbytes=lambda: float((x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0)
                                  + (aux.numel() if aux is not None else 0)) * x.element_size()))
        return y
    if stats_out is not None and aux is None and not transposed:
        slots = _conv_stats_slots(dt, d, n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, in_act, out_act)
        if slots:
            part = torch.empty(n * slots * cout * 2, dtype=torch.float32, device=x.device)
            LaunchProfiler.run('conv_patch', flops, lambda: L.check(
                L.lib().s2e_conv2d_stats(dt, _p(x), _p(wp), _p(bias), _p(residual), _p(y), C.byref(d), _p(part), _stream()), 's2e_conv2d_stats'),
                tag=lambda: 'F n%d %dx%d c%d->%d k%d s%d +stats' % (n, hi, wi, cin, cout, kh, stride),
                nbytes=lambda: float((x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0)) * x.element_size()))
            ws = torch.empty(n * cout * 2, dtype=torch.float64, device=x.device)
            stats = torch.empty(n, cout, 2, dtype=torch.float32, device=x.device)
            LaunchProfiler.run('in_stats', 0.0, lambda: L.check(
                L.lib().s2e_in_stats_from_partials(_p(part), n, slots, cout, ho * wo, IN_EPS, _p(ws), _p(stats), _stream()),
                's2e_in_stats_from_partials'), nbytes=float(part.numel() * 4))
            stats_out.append(stats)
            return y
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
    LaunchProfiler.run(lambda: _CONV_FAMILY[L.lib().s2e_conv2d_kernel_kind(dt, C.byref(d))], flops, lambda: L.check(
        L.lib().s2e_conv2d(dt, _p(x), _p(wp), _p(bias), _p(residual), _p(aux), _p(y), C.byref(d), _p(ws), wsb,
                           _stream()), 's2e_conv2d'),
        tag=lambda: '%s n%d %dx%d c%d->%d k%d s%d' % ('D' if transposed else 'F', n, hi, wi, cin, cout, kh, stride),
        # algorithmic bytes: every operand once (x, packed w, y, + residual / mask tensor when present)
        nbytes=lambda: float((x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0)
                              + (aux.numel() if aux is not None else 0)) * x.element_size()))
    return y


def conv2d_wgrad_raw(x, gy, kh, kw, stride, pad, in_act=ACT_NONE, want_bias=False, dbias_out=None, dw_out=None, gy_shared=False,
                     defer_ok=False):
    """-> (dw, db): dw (Cout, KH*KW*Cin) fp32 in packed order; db (Cout) fp32 or None.  Both live in one
    zero-filled buffer (ZeroPool scratch when the bias gradient is not returned).  dbias_out: an fp32 (Cout) tensor to ACCUMULATE the bias
    gradient into instead (e.g. the parameter's slice of the gradient arena); then db is None.
    dw_out: an fp32 (Cout, KH*KW*Cin) row-major tensor to ACCUMULATE the weight gradient into instead of a fresh zeroed buffer
    (the gradient of a parameter stored channels-last: _cl_rows(p.grad)); returned as dw.
    Inside a trainer step the patch-resident 3x3 shapes are QUEUED (GradSink.push_wgrad): dw / dbias_out then receive the sums at
    the step's next flush.  gy_shared: gy is also handed on as another tensor's gradient (see push_wgrad).
    defer_ok: the caller reads dw / db only through jobs queued in the same GradSink (or not at all before the flush: an arena slice):
    a generic shape may then be queued too (GradSink.push_gwg: every generic weight gradient of a backward as one launch)."""
    _need(x, gy, dbias_out, dw_out)
    n, hi, wi, cin = x.shape
    _, ho, wo, cout = gy.shape
    k = kh * kw * cin
    own_b = want_bias and dbias_out is None
    if dw_out is not None:
        if tuple(dw_out.shape) != (cout, k) or dw_out.dtype != torch.float32 or not dw_out.is_contiguous():
            raise ValueError('conv2d_wgrad_raw: dw_out must be a contiguous fp32 (%d, %d) tensor' % (cout, k))
        dw, db = dw_out, (torch.zeros(cout, dtype=torch.float32, device=x.device) if own_b else None)
    elif own_b:                                          # db goes back to autograd (may become a .grad): never pooled
        buf = torch.zeros(cout * k + cout, dtype=torch.float32, device=x.device)
        dw, db = buf[:cout * k].view(cout, k), buf[cout * k:]
    else:
        dw, db = ZeroPool.take(cout * k, torch.float32, x.device).view(cout, k), None
    dbp = db if own_b else dbias_out
    if (kh == 3 and kw == 3 and stride == 1 and pad == 1 and in_act == ACT_NONE and not own_b and ho == hi and wo == wi
            and GradSink.push_wgrad(x, gy, dw, dbp, gy_shared=gy_shared)):
        return dw, db                                        # accumulated at the step's next flush, with every other queued layer
    if defer_ok and not own_b and GradSink.push_gwg(x, gy, dw, dbp, (n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, 0, in_act, ACT_NONE, AUX_NONE),
                                                    gy_shared=gy_shared):
        return dw, db
    d, wsb = _conv_plan(True, _dt(x), n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, 0, in_act, ACT_NONE, AUX_NONE)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
    LaunchProfiler.run(lambda: _WGRAD_FAMILY[L.lib().s2e_conv2d_wgrad_kernel_kind(_dt(x), C.byref(d))],
                       2.0 * n * ho * wo * cin * cout * kh * kw, lambda: L.check(
        L.lib().s2e_conv2d_wgrad(_dt(x), _p(x), _p(gy), _p(dw), _p(dbp), C.byref(d), _p(ws), wsb, _stream()),
        's2e_conv2d_wgrad'),
        tag=lambda: 'W n%d %dx%d c%d->%d k%d s%d' % (n, hi, wi, cin, cout, kh, stride),
        nbytes=lambda: float((x.numel() + gy.numel()) * x.element_size() + dw.numel() * 4))
    return dw, db


def unpack_weight_grad_into(dwp, dst, cout, cin, kh, kw, cin_pad, accumulate=True):
    if accumulate and GradSink.push(dwp, dst, cout, cin, kh * kw, cin_pad):
        return
    L.check(L.lib().s2e_unpack_weight_grad(_p(dwp), _p(dst), cout, cin, kh, kw, cin_pad, int(accumulate), _stream()),
            's2e_unpack_weight_grad')


def _unpack_dw(dw, cout, cin, kh, kw, cin_pad):
    """(Cout, KH*KW*cin_pad) packed fp32 gradient -> (Cout, Cin, KH, KW) view."""
    return dw.view(cout, kh, kw, cin_pad)[..., :cin].permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ conv2d

class LivePrefix:
    """Batches of which only the first `n` samples carry a gradient.  In the G step netD runs on [fake | real] and the
    real half only serves as the (detached) target of the feature-matching loss: its gradient is exactly zero all the way
    down.  A conv recorded inside `with LivePrefix.of(n)` computes its DATA gradient for the first n samples only and
    zero-fills the rest -- the same numbers for half the MFMA work.  netD gates its outputs (live_prefix_gate) so that the
    premise holds whatever the caller does with them."""
    n = None

    class of:
        def __init__(self, n):
            self.n = n

        def __enter__(self):
            self.prev, LivePrefix.n = LivePrefix.n, self.n

        def __exit__(self, *exc):
            LivePrefix.n = self.prev


def _live_tail_buffer(x, live):
    """A gradient buffer shaped like x whose samples live.. are ZERO, for a data gradient that only writes samples ..live.
    Inside a trainer step the i-th such request of a step gets the i-th PERSISTENT buffer of the step's pool (same sequence
    every step): its tail was zeroed when it was made and nothing writes there -- the data-gradient kernel fills the head, the
    consumers (FeatTapFn, the IN backward) read it or accumulate into the head only -- so the ten zero-fill launches of a G
    step's discriminator backward disappear.  Stand-alone: a fresh tensor and one fill."""
    pool = ZeroPool.active()
    if pool is not None:
        key = (pool.key, pool.tail_i)
        pool.tail_i += 1
        ent = pool.tails.get(key)
        if ent is not None and ent[0] == live and ent[1].shape == x.shape and ent[1].dtype == x.dtype and ent[1].device == x.device:
            return ent[1]
        if not pool.frozen:                                  # (a captured graph must not start using new persistent memory)
            t = torch.zeros_like(x)
            pool.tails[key] = (live, t)
            return t
    gx = torch.empty_like(x)
    gx[live:].zero_()
    return gx


class _LivePrefixGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, n):
        ctx.n = n
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        pool = ZeroPool.active()
        if pool is not None and any(live == ctx.n and t.data_ptr() == g.data_ptr() and t.shape == g.shape for live, t in pool.tails.values()):
            return g, None                                   # one of the step's zero-tailed buffers (_live_tail_buffer): nothing to do
        g = g.clone()
        g[ctx.n:].zero_()
        return g, None


def live_prefix_gate(h, n):
    """Identity whose backward zeroes the gradient of samples n.. (see LivePrefix)."""
    return _LivePrefixGate.apply(h, n)


class Conv2dFn(torch.autograd.Function):
    """y = out_act(conv(in_act(x), W) + b + residual) on NHWC tensors.  x may carry more channels
    than W has input channels (structural zero padding).  With (u, v, sigma) given, W = weight/sigma
    (spectral norm): the division happens inside the pack kernel and the gradient returned for
    `weight` is the one w.r.t. weight_orig, through sigma."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad, in_act, out_act, u, v, sigma, stats_out=None):
        n, hi, wi, cx = x.shape
        cout, cin, kh, kw = weight.shape
        if cx < cin:
            raise ValueError('input has %d channels, weight expects %d' % (cx, cin))
        ho = (hi + 2 * pad - kh) // stride + 1
        wo = (wi + 2 * pad - kw) // stride + 1
        plan = packing.current()
        pm = plane_mode(x.dtype, n, hi, wi, cx, ho, wo, cout, kh, kw, stride, pad, False, in_act, out_act, AUX_NONE) if cx == cin else 0
        wp = packed_weight(weight, x.dtype, cx, False, sigma, plan, plane=pm > 0)
        ctx.plan, ctx.plan_gen = plan, (plan.generation if plan is not None else None)
        b = None if bias is None else bias.detach().float().contiguous()
        y = conv2d_raw(x, wp, b, residual, None, (ho, wo, cout), kh, kw, stride, pad, False, in_act, out_act, stats_out=stats_out, plane=pm > 0)
        ctx.cfg = (stride, pad, in_act, out_act, bias is not None, residual is not None)
        ctx.live = LivePrefix.n
        ctx.wdst = _grad_dst(weight)                       # direct accumulation targets (or None)
        ctx.bdst = _grad_dst(bias) if bias is not None else None
        ctx.save_for_backward(x, weight, y if out_act != ACT_NONE else None, u, v, sigma)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, u, v, sigma = ctx.saved_tensors
        stride, pad, in_act, out_act, has_bias, has_res = ctx.cfg
        n, hi, wi, cx = x.shape
        cout, cin, kh, kw = weight.shape
        g = gy.contiguous()
        live = ctx.live
        if (live is not None and 0 < live < n and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1]
                and not (has_bias and ctx.needs_input_grad[2]) and not (has_res and ctx.needs_input_grad[3])
                and out_act in (ACT_NONE, ACT_LRELU)):
            # only the first `live` samples carry a gradient (LivePrefix): data gradient of that prefix, zeros behind it
            gl = g[:live]
            if out_act == ACT_LRELU:
                g2 = torch.empty_like(gl)
                L.check(L.lib().s2e_lrelu_bwd(_dt(gl), _p(gl), _p(y), _p(g2), gl.numel(), _stream()), 's2e_lrelu_bwd')
                gl = g2
            gx = _live_tail_buffer(x, live)                    # (samples live.. are zero already)
            am = AUX_LRELU_GRAD if in_act == ACT_LRELU else AUX_NONE
            pm = plane_mode(x.dtype, live, gl.shape[1], gl.shape[2], cout, hi, wi, cx, kh, kw, stride, pad, True, ACT_NONE, ACT_NONE, am) if cx == cin else 0
            wpt = packed_weight(weight, x.dtype, cx, True, sigma, ctx.plan, ctx.plan_gen, plane=pm > 0)
            conv2d_raw(gl, wpt, None, None, x[:live] if in_act == ACT_LRELU else None, (hi, wi, cx), kh, kw, stride, pad,
                       True, ACT_NONE, ACT_NONE, am, out=gx[:live], plane=pm > 0)
            return gx, None, None, None, None, None, None, None, None, None, None, None
        if out_act == ACT_TANH:
            g2 = torch.empty_like(g)
            L.check(L.lib().s2e_tanh_bwd(_dt(g), _p(g), _p(y), _p(g2), g.numel(), _stream()), 's2e_tanh_bwd')
            g = g2
        elif out_act == ACT_LRELU:
            g2 = torch.empty_like(g)
            L.check(L.lib().s2e_lrelu_bwd(_dt(g), _p(g), _p(y), _p(g2), g.numel(), _stream()), 's2e_lrelu_bwd')
            g = g2
        gx = gw = gb = gres = None
        if ctx.needs_input_grad[0]:
            am = AUX_LRELU_GRAD if in_act == ACT_LRELU else AUX_NONE
            pm = plane_mode(x.dtype, n, g.shape[1], g.shape[2], cout, hi, wi, cx, kh, kw, stride, pad, True, ACT_NONE, ACT_NONE, am) if cx == cin else 0
            wpt = packed_weight(weight, x.dtype, cx, True, sigma, ctx.plan, ctx.plan_gen, plane=pm > 0)
            gx = conv2d_raw(g, wpt, None, None, x if in_act == ACT_LRELU else None, (hi, wi, cx), kh, kw, stride, pad,
                            True, ACT_NONE, ACT_NONE, am, plane=pm > 0)
        want_b = has_bias and ctx.needs_input_grad[2]
        wdst = ctx.wdst
        shared = bool(has_res and ctx.needs_input_grad[3])   # g goes on as the residual's gradient (and may be added to in place there)
        direct = ctx.needs_input_grad[1] and wdst is not None and cx == cin and cin % 8 == 0 and _cl_dense(wdst)
        if direct and sigma is not None and not GradSink.inplace_allowed(wdst):
            direct = False                                   # (the chain rule must ACCUMULATE here: packed scratch, below)
        if ctx.needs_input_grad[1] and sigma is not None and not direct and ctx.wdst is not None:
            ZeroPool.arena_touched(ctx.wdst)                 # (.grad now holds a chain-ruled part: no in-place rewrite before zero_grad)
        if direct:
            # the parameter's gradient lies in the packed order (channels-last arena, or any 1x1 conv; Cin % 8 == 0 -- a 1-channel
            # weight is "channels-last" too, but the in-place kernels work on 16-byte groups of one tap): the kernel accumulates
            # straight into it; spectral norm's chain rule is then applied in place (queued: one launch pair per step)
            _, gb = conv2d_wgrad_raw(x, g, kh, kw, stride, pad, in_act, want_b, ctx.bdst if want_b else None, dw_out=_cl_rows(wdst),
                                     gy_shared=shared, defer_ok=True)
            if sigma is not None:
                GradSink.push_inplace(_cl_rows(wdst), weight, u, v, sigma, cout, cin, kh * kw)
        elif ctx.needs_input_grad[1]:
            bdst = ctx.bdst if want_b else None
            if wdst is not None and not wdst.is_contiguous():
                wdst = None                                  # (a channels-last .grad fed a channel-padded input: through autograd)
            # (with a .grad to accumulate into, the packed dW is read by jobs queued in the step's GradSink only: it may be queued itself)
            dwp, gb = conv2d_wgrad_raw(x, g, kh, kw, stride, pad, in_act, want_b, bdst, gy_shared=shared, defer_ok=wdst is not None)
            w_oihw = weight.detach() if weight.is_contiguous() else weight.detach().contiguous()
            if sigma is None:
                if wdst is not None:
                    unpack_weight_grad_into(dwp, wdst, cout, cin, kh, kw, cx)
                else:
                    gw = _unpack_dw(dwp, cout, cin, kh, kw, cx)
            else:
                acc = wdst is not None
                if not (acc and GradSink.push(dwp, wdst, cout, cin, kh * kw, cx, w_oihw, u, v, sigma)):
                    out = wdst if acc else torch.empty(cout, cin, kh, kw, dtype=torch.float32, device=x.device)
                    dot = ZeroPool.take(1, torch.float32, x.device)
                    L.check(L.lib().s2e_sn_weight_grad(_p(dwp), _p(w_oihw), _p(u), _p(v), _p(sigma), _p(dot), _p(out),
                                                       cout, cin, kh, kw, cx, int(acc), _stream()), 's2e_sn_weight_grad')
                    gw = None if acc else out
        elif want_b:
            gb = colsum(g)
        if has_res and ctx.needs_input_grad[3]:
            gres = g
        return gx, gw, gb, gres, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, residual=None, stride=1, pad=0, in_act=ACT_NONE, out_act=ACT_NONE, sn=None, stats_out=None):
    """stats_out: see conv2d_raw -- a list that receives in_stats(result) when the conv kernel can produce it."""
    u, v, sigma = sn if sn is not None else (None, None, None)
    return Conv2dFn.apply(x, weight, bias, residual, stride, pad, in_act, out_act, u, v, sigma, stats_out)


def conv2d_m(x, conv, residual=None, stride=1, pad=0, in_act=ACT_NONE, out_act=ACT_NONE, stats_out=None):
    """conv2d on an nn.Conv2d parameter container (spectral-normed or not)."""
    from ..spectral import conv_params
    weight, bias, sn = conv_params(conv)
    return conv2d(x, weight, bias, residual, stride, pad, in_act, out_act, sn, stats_out)


# ------------------------------------------------------------------------------ the encoder's head
class FcHeadFn(torch.autograd.Function):
    """y = fc(LeakyReLU(x).view(M, -1)) for an NHWC feature map x (M,h,w,C) and an nn.Linear whose input features are torch's
    (c, y, x) flattening (reference models/networks/encoder.py:68-71): s2e_fc_head_fwd / _bwd.  The weight / bias gradients are
    accumulated into the parameters' .grad when that is an fp32 arena view (None then goes back to autograd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope):
        _need(x, weight, bias)
        m, h, w, c = x.shape
        n = weight.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        wsb = L.lib().s2e_fc_head_fwd_workspace_bytes(m, h * w, c, n)
        ws = torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=x.device)
        L.check(L.lib().s2e_fc_head_fwd(_dt(x), _p(x), _p(weight), _p(bias), _p(y), m, h * w, c, n, float(slope), _p(ws), wsb, _stream()),
                's2e_fc_head_fwd')
        ctx.slope = float(slope)
        ctx.wdst, ctx.bdst = _grad_dst(weight), _grad_dst(bias)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        m, h, w, c = x.shape
        n = weight.shape[0]
        g = gy.float().contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        wdst = ctx.wdst if (ctx.wdst is not None and ctx.wdst.is_contiguous()) else None
        dw = wdst if wdst is not None else (torch.zeros_like(weight) if ctx.needs_input_grad[1] else None)
        db = ctx.bdst if ctx.bdst is not None else (torch.zeros(n, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None)
        L.check(L.lib().s2e_fc_head_bwd(_dt(x), _p(x), _p(weight), _p(g), _p(dx), _p(dw), _p(db), m, h * w, c, n, ctx.slope, _stream()),
                's2e_fc_head_bwd')
        return dx, (None if wdst is not None else dw), (None if ctx.bdst is not None else db), None


def fc_head(x, weight, bias, slope=0.2):
    """-> (M, N) fp32, or None when the shape is outside the kernel's range (the caller then takes the convolution form)."""
    if (weight.dtype != torch.float32 or not weight.is_contiguous() or bias is None or weight.shape[1] != x.shape[1] * x.shape[2] * x.shape[3]
            or weight.shape[1] * 4 > 48 * 1024 or not L.lib().s2e_fc_head_supported(x.shape[0], weight.shape[0])):
        return None
    return FcHeadFn.apply(x.contiguous(), weight, bias, slope)
