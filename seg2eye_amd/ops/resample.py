"""Resampling and tensor plumbing: nearest 2x upsampling, bilinear resize, 3x3 stride-2 average pooling, the discriminator's
8-channel [one-hot | image] input and the split of its [fake | real] batch."""
import torch

from .. import _lib as L
from .core import LaunchProfiler, _dt, _need, _p, _single_channel, _stream
from .spade import onehot_nhwc_raw


# ------------------------------------------------------------------------------ resampling

class Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        n, h, w, c = x.shape
        y = torch.empty(n, 2 * h, 2 * w, c, dtype=x.dtype, device=x.device)
        LaunchProfiler.run('resample', 0.0, lambda: L.check(
            L.lib().s2e_upsample2x_fwd(_dt(x), _p(x), _p(y), n, h, w, c, _stream()), 's2e_upsample2x_fwd'),
            nbytes=float(5 * x.numel() * x.element_size()))
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        n, h2, w2, c = gy.shape
        gx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().s2e_upsample2x_bwd(_dt(gy), _p(gy), _p(gx), n, h2 // 2, w2 // 2, c, _stream()), 's2e_upsample2x_bwd')
        return gx


def upsample2x(x):
    return Upsample2xFn.apply(x)


class BilinearResizeFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear') of single-channel images (N,1,H,W) fp32 -> (N,h,w,1) NHWC in `dtype`
    (encoder.py:54-55)."""

    @staticmethod
    def forward(ctx, x, h, w, dtype):
        a = _single_channel(x.detach().float())
        _need(a)
        n, H, W = a.shape
        y = torch.empty(n, h, w, 1, dtype=dtype, device=a.device)
        L.check(L.lib().s2e_bilinear_resize_fwd(_dt(y), _p(a), _p(y), n, H, W, h, w, _stream()), 's2e_bilinear_resize_fwd')
        ctx.shape = (tuple(x.shape), H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, H, W = ctx.shape
        gy = gy.contiguous()
        n, h, w, _ = gy.shape
        gx = torch.zeros(n, H, W, dtype=torch.float32, device=gy.device)
        L.check(L.lib().s2e_bilinear_resize_bwd(_dt(gy), _p(gy), _p(gx), n, H, W, h, w, _stream()), 's2e_bilinear_resize_bwd')
        return gx.view(shape), None, None, None


def bilinear_resize(x, h, w, dtype):
    return BilinearResizeFn.apply(x, h, w, dtype)


class AvgPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        n, h, w, c = x.shape
        y = torch.empty(n, (h + 1) // 2, (w + 1) // 2, c, dtype=x.dtype, device=x.device)
        L.check(L.lib().s2e_avgpool3x3s2_fwd(_dt(x), _p(x), _p(y), n, h, w, c, _stream()), 's2e_avgpool3x3s2_fwd')
        ctx.hw = (h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        h, w = ctx.hw
        n, _, _, c = gy.shape
        gx = torch.empty(n, h, w, c, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().s2e_avgpool3x3s2_bwd(_dt(gy), _p(gy), _p(gx), n, h, w, c, _stream()), 's2e_avgpool3x3s2_bwd')
        return gx


def avgpool3x3s2(x):
    return AvgPool3x3s2Fn.apply(x)


class SegImageConcatFn(torch.autograd.Function):
    """cat([one_hot(label), image], channel) as an NHWC tensor zero-padded to `cpad` channels
    (pix2pix_model.py:328-336).  Differentiable w.r.t. the image only."""

    @staticmethod
    def forward(ctx, label, img, ncls, cpad):
        n, H, W = label.shape
        ctx.ncls = ncls
        return onehot_nhwc_raw(label, img.contiguous(), H, W, ncls, cpad, img.dtype)

    @staticmethod
    def backward(ctx, g):
        return None, g[..., ctx.ncls].contiguous(), None, None


def seg_image_concat(label, img, ncls=4, cpad=8):
    """label (N,H,W) uint8, img (N,H,W) -> (N,H,W,cpad)."""
    return SegImageConcatFn.apply(label, img, ncls, cpad)


class DInputFn(torch.autograd.Function):
    """The discriminator's input  cat_batch([cat_ch(one_hot(label), fake); cat_ch(one_hot(label), real)])  (pix2pix_model.py:
    328-342) as ONE (2N,H,W,cpad) NHWC tensor written by two launches -- one per half, straight from the label map and the two
    single-channel image batches: no concatenated image batch, no doubled label map.  Differentiable w.r.t. `fake` only; its
    gradient is channel `ncls` of the first half (one strided copy instead of select_backward's zero-fill + copy)."""

    @staticmethod
    def forward(ctx, label, fake, real, ncls, cpad):
        n, H, W = label.shape
        f, r = fake.reshape(n, H, W), real.reshape(n, H, W).to(fake.dtype)
        f, r = (f if f.is_contiguous() else f.contiguous()), (r if r.is_contiguous() else r.contiguous())
        _need(label, f, r)
        out = torch.empty(2 * n, H, W, cpad, dtype=fake.dtype, device=label.device)
        for half, img in ((out[:n], f), (out[n:], r)):
            L.check(L.lib().s2e_onehot_nhwc(_dt(out), _p(label), _p(img), _p(half), n, H, W, H, W, ncls, cpad, _stream()),
                    's2e_onehot_nhwc')
        ctx.ncls, ctx.shape = ncls, fake.shape
        return out

    @staticmethod
    def backward(ctx, g):
        n = g.shape[0] // 2
        return None, g[:n, :, :, ctx.ncls].contiguous().view(ctx.shape), None, None, None


def d_input(label, fake, real, ncls=4, cpad=8):
    """label (N,H,W) uint8, fake / real (N,1,H,W) or (N,H,W) -> (2N,H,W,cpad): see DInputFn."""
    return DInputFn.apply(label, fake, real, ncls, cpad)


class SplitHalvesFn(torch.autograd.Function):
    """t -> (t[:n], t[n:]), n = half the batch (Pix2PixModel.divide_pred).  Two plain slices cost a zero-filled full tensor and
    a copy EACH on the way back, plus the add that joins them; here the backward fills one tensor with the two halves (or
    zeros where a half got no gradient)."""

    @staticmethod
    def forward(ctx, t):
        ctx.set_materialize_grads(False)
        n = t.shape[0] // 2
        ctx.n = n
        ctx.like = t.detach()
        return t[:n], t[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None
        out = torch.empty_like(ctx.like)
        for dst, g in ((out[:ctx.n], ga), (out[ctx.n:], gb)):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        return out


def split_halves(t):
    return SplitHalvesFn.apply(t)
