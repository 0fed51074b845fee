"""SPADE+Style residual block (reference models/networks/architecture.py:13-62)."""
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from .. import ops
from ..spectral import sn_begin
from .normalization import SPADE_STYLE_Block, SegMap, spade_stats


class SPADE_STYLE_ResnetBlock(nn.Module):
    """x_s = conv_s(SSB_s(x)) if fin != fout else x;  dx = conv_0(lrelu(SSB_0(x)));
    dx = conv_1(lrelu(SSB_1(dx)));  out = x_s + dx.
    HIP schedule: IN statistics of x are computed once and shared by norm_0 and norm_s; the
    LeakyReLU is fused into the modulation kernel; bias and the residual add are fused into the
    conv epilogue.  x, out: (N,h,w,C) NHWC."""

    def __init__(self, fin, fout, opt):
        super().__init__()
        fmid = min(fin, fout)
        self.learned_shortcut = fin != fout
        wrap = spectral_norm if 'spectral' in opt.norm_G else (lambda conv: conv)
        # (name, in, out, kernel, bias) / (name, channels); registered in the order of the reference's state_dict:
        # conv_0, conv_1, [conv_s], norm_0, norm_1, [norm_s]
        convs = [('conv_0', fin, fmid, 3, True), ('conv_1', fmid, fout, 3, True)]
        norms = [('norm_0', fin), ('norm_1', fmid)]
        if self.learned_shortcut:
            convs.append(('conv_s', fin, fout, 1, False))
            norms.append(('norm_s', fin))
        for name, cin, cout, k, bias in convs:
            setattr(self, name, wrap(nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=bias)))
        for name, c in norms:
            setattr(self, name, SPADE_STYLE_Block(c, opt))

    def input_stats(self, x, replication=1):
        """Statistics for norm_0 / norm_s of this block from x BEFORE the generator's nearest 2x upsampling
        (replication=4): the same numbers from a quarter of the bytes.  Pass the result to forward(stats=...).
        When x is the output of a block whose last conv produced the statistics in its epilogue (`_s2e_in_stats`, set by
        forward below), they are taken from there: no pass over x at all."""
        ready = getattr(x, '_s2e_in_stats', None)
        if ready is not None and self.norm_0.spade.kind == 'instance':
            return ready
        return spade_stats(x, [self.norm_0.spade] + ([self.norm_s.spade] if self.learned_shortcut else []), replication)

    def _convs(self, h0, x_s, seg, latent_style):
        """dx = conv_0(h0); out = conv_1(lrelu(SSB_1(dx))) + x_s.  With InstanceNorm SPADE the statistics norm_1 needs of dx,
        and the statistics the NEXT block needs of out, come out of the two convs' epilogues when their kernel has that
        epilogue (ops.conv2d_raw stats_out) -- otherwise norm_1 / the next block run the statistics pass."""
        fused = self.norm_1.spade.kind == 'instance'
        s_dx, s_out = ([], []) if fused else (None, None)
        dx = ops.conv2d_m(h0, self.conv_0, None, 1, 1, stats_out=s_dx)
        out = ops.conv2d_m(self.norm_1(dx, seg, latent_style, s_dx[0] if s_dx else None, lrelu=True), self.conv_1, x_s, 1, 1, stats_out=s_out)
        if s_out:
            out._s2e_in_stats = s_out[0]
        return out

    def forward(self, x, seg, latent_style, stats=None, up=False):
        """up (not in the reference): x is the tensor BEFORE the generator's nearest 2x upsampling and `stats` were taken from
        it (input_stats(x, 4)).  In a block with a learned shortcut x feeds norm_0 and norm_s only, whose fused launches read it
        at (y/2, x/2) and whose backward returns the gradient w.r.t. it (the 2 x 2 sums): neither the upsampled tensor -- 4x the
        bytes, written once and read twice per pass -- nor its gradient ever exists.  Otherwise the block upsamples first
        (ops.upsample2x), as generator.py:77-92 does."""
        seg = SegMap.of(seg)
        sn_begin(self)                  # no-op inside a generator (its forward already stepped the bank)
        fold = (up and self.learned_shortcut and stats is not None and self.norm_0.spade.kind != 'batch'
                and self.norm_0.takes_fold(x) and self.norm_s.takes_fold(x))
        if up and not fold:
            x = ops.upsample2x(x)
        if fold:
            if torch.is_grad_enabled() and x.requires_grad:
                h0, x = self.norm_0(x, seg, latent_style, stats, lrelu=True, relay=True, up=True)
            else:
                h0 = self.norm_0(x, seg, latent_style, stats, lrelu=True, up=True)
            x_s = ops.conv2d_m(self.norm_s(x, seg, latent_style, stats, lrelu=False, up=True), self.conv_s)
            return self._convs(h0, x_s, seg, latent_style)
        if stats is None:
            stats = self.input_stats(x)
        # x has two consumers (norm_0 and norm_s, or norm_0 and the residual).  With gradients on, the second one hangs off
        # an alias of x that norm_0 hands out, so its gradient reaches norm_0's backward and is accumulated there in place
        # instead of autograd adding two full-size tensors.
        if torch.is_grad_enabled() and x.requires_grad:
            h0, x = self.norm_0(x, seg, latent_style, stats, lrelu=True, relay=True)
        else:
            h0 = self.norm_0(x, seg, latent_style, stats, lrelu=True)
        if self.learned_shortcut:
            x_s = ops.conv2d_m(self.norm_s(x, seg, latent_style, stats, lrelu=False), self.conv_s)
        else:
            x_s = x
        return self._convs(h0, x_s, seg, latent_style)
