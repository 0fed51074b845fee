"""Loss reductions (hinge / L1 / feature matching with in-place gradient injection), the flat Adam step, and the OpenEDS validation
metric kernels."""
import torch

from .. import _lib as L
from .._lib import LOSS_L1
from .core import LaunchProfiler, ZeroPool, _dt, _need, _p, _single_channel, _stream
from .conv import _live_tail_buffer


# ------------------------------------------------------------------------------ losses

def _loss_slot(device, pooled):
    """A zeroed fp32 scalar for s2e_loss_reduce to accumulate into.  Pool memory is recycled by the trainer's next step: only
    for terms that are consumed inside the step (see loss_sum)."""
    if pooled and ZeroPool.active() is not None:
        return ZeroPool.take(1, torch.float32, device).view(())
    return torch.zeros((), dtype=torch.float32, device=device)


class LossSumFn(torch.autograd.Function):
    """scale * sum_i f(a_i, b_i) as a 0-dim fp32 tensor (see s2e_loss_reduce for f)."""

    @staticmethod
    def forward(ctx, a, b, mode, scale, pooled=False):
        _need(a, b)
        out = _loss_slot(a.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(a), mode, _p(a), _p(b), a.numel(), float(scale), _p(out), _stream()),
                's2e_loss_reduce')
        ctx.cfg = (mode, float(scale))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        mode, scale = ctx.cfg
        gs = gout.detach().float().contiguous()
        da = torch.empty_like(a)
        L.check(L.lib().s2e_loss_grad(_dt(a), mode, _p(a), _p(b), a.numel(), scale, _p(gs), _p(da), 0, _stream()),
                's2e_loss_grad')
        return da, None, None, None, None


class HalfLossFn(torch.autograd.Function):
    """scale * sum_i f(t_i) over ONE half of a [fake | real] batch t (2N, ...), NHWC-contiguous; the other half gets no gradient.
    == loss_sum(t[:N] or t[N:], ...) without the slice: the backward writes the element-wise gradient into the head of a
    zero-tailed buffer (first half, inside a trainer step: _live_tail_buffer -- one launch, and the LivePrefix gate behind it
    recognises the buffer) instead of slice_backward's zero-fill + copy."""

    @staticmethod
    def forward(ctx, t, second, mode, scale, pooled):
        _need(t)
        n = t.shape[0] // 2
        half = t[n:] if second else t[:n]
        out = _loss_slot(t.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(t), mode, _p(half), None, half.numel(), float(scale), _p(out), _stream()), 's2e_loss_reduce')
        ctx.cfg = (bool(second), mode, float(scale), n)
        ctx.save_for_backward(t)
        return out

    @staticmethod
    def backward(ctx, gout):
        t, = ctx.saved_tensors
        second, mode, scale, n = ctx.cfg
        gs = gout.detach().float().contiguous()
        if second:
            gt = torch.empty_like(t)
            gt[:n].zero_()
            dst, src = gt[n:], t[n:]
        else:
            gt = _live_tail_buffer(t, n)
            dst, src = gt[:n], t[:n]
        L.check(L.lib().s2e_loss_grad(_dt(t), mode, _p(src), None, src.numel(), scale, _p(gs), _p(dst), 0, _stream()), 's2e_loss_grad')
        return gt, None, None, None, None


class PairLossFn(torch.autograd.Function):
    """(scale * sum f_a(t[:N]), scale * sum f_b(t[N:])) for a [fake | real] batch t: the discriminator's two hinge terms from
    the undivided prediction; the backward fills ONE gradient tensor with two launches (no slice_backward, no add)."""

    @staticmethod
    def forward(ctx, t, mode_a, mode_b, scale, pooled):
        _need(t)
        n = t.shape[0] // 2
        outs = []
        for half, mode in ((t[:n], mode_a), (t[n:], mode_b)):
            out = _loss_slot(t.device, pooled)
            L.check(L.lib().s2e_loss_reduce(_dt(t), mode, _p(half), None, half.numel(), float(scale), _p(out), _stream()), 's2e_loss_reduce')
            outs.append(out)
        ctx.cfg = (mode_a, mode_b, float(scale), n)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(t)
        return tuple(outs)

    @staticmethod
    def backward(ctx, ga, gb):
        t, = ctx.saved_tensors
        mode_a, mode_b, scale, n = ctx.cfg
        if ga is None and gb is None:
            return None, None, None, None, None
        gt = torch.empty_like(t)
        for dst, src, mode, g in ((gt[:n], t[:n], mode_a, ga), (gt[n:], t[n:], mode_b, gb)):
            if g is None:
                dst.zero_()
                continue
            gs = g.detach().float().contiguous()
            L.check(L.lib().s2e_loss_grad(_dt(t), mode, _p(src), None, src.numel(), scale, _p(gs), _p(dst), 0, _stream()), 's2e_loss_grad')
        return gt, None, None, None, None


def half_loss(t, second, mode, scale, pooled=False):
    return HalfLossFn.apply(t, second, mode, scale, pooled)


def pair_loss(t, mode_a, mode_b, scale, pooled=False):
    return PairLossFn.apply(t, mode_a, mode_b, scale, pooled)


def loss_sum(a, b, mode, scale, pooled=False):
    """pooled: the caller only COMBINES the result with other terms inside the step (a sum over scales, a stack) and never
    hands it out: the accumulator may then be a slice of the step's zero pool instead of its own zero-fill launch."""
    return LossSumFn.apply(a, b, mode, scale, pooled)


class FeatTapFn(torch.autograd.Function):
    """Identity on a discriminator feature map h = [fake | real] (2N,H,W,C) that also yields the GAN feature-
    matching term  scale * sum |h[:N] - h[N:].detach()|  (pix2pix_model.py:231-241 of the reference).

    Why not slice-then-loss: the slice's backward materialises a zero (2N,...) tensor, copies the half in and
    autograd then ADDS it to the gradient arriving from the next layer -- three passes over every feature map
    (~0.65 ms per G step).  Here the next layer's gradient arrives first (this node sits on the only path to
    it) and the L1 gradient is accumulated into its fake half in place by s2e_loss_grad(accumulate=1)."""

    @staticmethod
    def forward(ctx, h, scale, pooled=False):
        _need(h)
        n = h.shape[0] // 2
        a, b = h[:n], h[n:]
        out = _loss_slot(h.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(h), LOSS_L1, _p(a), _p(b), a.numel(), float(scale), _p(out), _stream()),
                's2e_loss_reduce')
        ctx.scale = float(scale)
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return h.view_as(h), out

    @staticmethod
    def backward(ctx, gh, gloss):
        h, = ctx.saved_tensors
        n = h.shape[0] // 2
        if gloss is None:
            return gh, None, None
        if gh is None:
            gh = torch.zeros_like(h)
        elif not gh.is_contiguous():
            gh = gh.contiguous()
        a, b, ga = h[:n], h[n:], gh[:n]
        gs = gloss.detach().float().contiguous()
        L.check(L.lib().s2e_loss_grad(_dt(h), LOSS_L1, _p(a), _p(b), a.numel(), ctx.scale, _p(gs), _p(ga), 1, _stream()),
                's2e_loss_grad')
        return gh, None, None


def feat_tap(h, scale, pooled=False):
    """-> (h, term): see FeatTapFn.  pooled: as in loss_sum."""
    return FeatTapFn.apply(h, scale, pooled)


# ------------------------------------------------------------------------------ optimizer

def adam_flat_step(p, g, m, v, hyper, skips_m=False):
    """One torch.optim.Adam step over flat fp32 arenas (pix2pix_model.py:92-110 semantics).
    hyper: 7-float DEVICE tensor {lr, beta1, beta2, eps, completed steps, grad_scale, weight_decay}.
    skips_m: the caller knows beta1 == 0 and weight_decay == 0 (the kernel then leaves m alone): only the profiler's byte count uses it."""
    _need(p, g, m, v, hyper)
    LaunchProfiler.run('adam', 0.0, lambda: L.check(
        L.lib().s2e_adam_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _stream()), 's2e_adam_flat'),
        # read p, g, m, v + write p, m, v = 28 B per parameter; without the first moment 20 B
        nbytes=float((5 if skips_m else 7) * 4 * p.numel()))


def openeds_error(produced, target):
    """Per-image OpenEDS error of two batches in [-1, 1] (models/networks/loss.py:135-155 `calculate_mse_for_tensors`):
    both mapped to 0..255 with the reference's int truncation, then sqrt(sum d^2) / (H*W).  -> fp32 (N,), no gradient."""
    a, b = _single_channel(produced.detach()), _single_channel(target.detach().to(produced.dtype))
    _need(a, b)
    n, h, w = a.shape
    err = torch.empty(n, dtype=torch.float32, device=a.device)
    L.check(L.lib().s2e_openeds_error(_dt(a), _p(a), _p(b), n, h, w, _p(err), _stream()), 's2e_openeds_error')
    return err


def openeds_error_u8(produced, target):
    """The same on uint8 images that already are 0..255 (loss.py:116-133 `calculate_mse_for_images`)."""
    a, b = _single_channel(produced), _single_channel(target)
    if a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise ValueError('uint8 images expected')
    _need(a, b)
    n, h, w = a.shape
    err = torch.empty(n, dtype=torch.float32, device=a.device)
    L.check(L.lib().s2e_openeds_error_u8(_p(a), _p(b), n, h, w, _p(err), _stream()), 's2e_openeds_error_u8')
    return err


def resize_to255(x, w=400, h=640):
    """Bilinear resize (cv2.INTER_LINEAR rule) of single-channel [-1, 1] images to (h, w), then 0..255 with int truncation
    (data/postprocessor.py:92-107 `to_255resized_imagebatch`).  -> uint8 (N,1,h,w)."""
    a = _single_channel(x.detach())
    _need(a)
    n, hi, wi = a.shape
    out = torch.empty(n, 1, h, w, dtype=torch.uint8, device=a.device)
    L.check(L.lib().s2e_resize_to255(_dt(a), _p(a), n, hi, wi, _p(out), h, w, _stream()), 's2e_resize_to255')
    return out
