"""Multi-scale PatchGAN discriminator (reference models/networks/discriminator.py:14-116)."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_NONE, ACT_LRELU
from .. import packing
from ..spectral import sn_begin
from .base_network import BaseNetwork, compute_dtype_of
from .normalization import apply_nonspade_norm, get_nonspade_norm_layer

import os
_TWO_STREAMS = os.environ.get('S2E_D_STREAMS', '0') == '1'      # round 6 experiment: the two scales of netD on two streams (measured neutral: off)

D_CPAD = 8          # the 5-channel input cat([one-hot seg, image]) is stored as 8 NHWC channels (16-B vectors)


def to_d_input(x, dtype):
    """(2N, label_nc+output_nc, H, W) float tensor of the reference API -> (2N,H,W,8) NHWC.
    Slow path (torch copies); Pix2PixModel builds the same tensor with ops.seg_image_concat."""
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, D_CPAD, dtype=dtype, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1).to(dtype)
    return out


class NLayerDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--n_layers_D', type=int, default=4, help='# layers in each discriminator')
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        kw = 4
        self.padw = int(np.ceil((kw - 1.0) / 2))
        nf = opt.ndf
        input_nc = opt.label_nc + opt.output_nc
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_D)
        sequence = [[nn.Conv2d(input_nc, nf, kernel_size=kw, stride=2, padding=self.padw), nn.LeakyReLU(0.2, False)]]
        self.strides = [2]
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            self.strides.append(stride)
            sequence += [[norm_layer(nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=stride, padding=self.padw)),
                          nn.LeakyReLU(0.2, False)]]
        sequence += [[nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=self.padw)]]
        self.strides.append(1)
        for n in range(len(sequence)):
            self.add_module('model' + str(n), nn.Sequential(*sequence[n]))
        self.n_groups = len(sequence)

    def forward_nhwc(self, x, feat_terms=None, feat_lambda=0.0):
        """x: (M,H,W,8).  Returns the NHWC outputs of model0..model{n} (after their activations).
        feat_terms: a list -> x is [fake | real] over the batch and every intermediate output is passed through
        ops.feat_tap: its feature-matching term (x feat_lambda / numel of the fake half) is appended to the list,
        its gradient is injected on the way back, and the returned intermediate features are detached."""
        sn_begin(self)                  # no-op when MultiscaleDiscriminator.forward already stepped
        feats = []

        def tap(h):
            if feat_terms is None:
                feats.append(h)
                return h
            h, term = ops.feat_tap(h, feat_lambda / (h.numel() // 2), pooled=True)     # (the terms are stacked and summed by the caller)
            feat_terms.append(term)
            feats.append(h.detach())
            return h
        first = self.model0[0]
        h = ops.conv2d(x, first.weight, first.bias, None, 2, self.padw, ACT_NONE, ACT_LRELU)   # conv + LeakyReLU, one launch
        h = tap(h)
        for n in range(1, self.n_groups - 1):
            blk = getattr(self, 'model%d' % n)[0]
            if isinstance(blk, nn.Sequential):                     # SN conv (bias removed) -> InstanceNorm -> LeakyReLU
                conv = blk[0]
                h = ops.conv2d_m(h, conv, None, self.strides[n], self.padw)
                h = apply_nonspade_norm(h, blk[1], lrelu=True)
            else:                                                  # norm_D without a norm layer: conv -> LeakyReLU
                h = ops.conv2d_m(h, blk, None, self.strides[n], self.padw, ACT_NONE, ACT_LRELU)
            h = tap(h)
        last = getattr(self, 'model%d' % (self.n_groups - 1))[0]
        h = ops.conv2d(h, last.weight, last.bias, None, 1, self.padw)
        if ops.LivePrefix.n is not None:
            h = ops.live_prefix_gate(h, ops.LivePrefix.n)          # the prediction for the real half: no gradient either
        feats.append(h)
        return feats

    def forward(self, input):
        feats = self.forward_nhwc(input)
        outs = [f.permute(0, 3, 1, 2) for f in feats]
        return outs if not self.opt.no_ganFeat_loss else outs[-1]


class MultiscaleDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--netD_subarch', type=str, default='n_layer', help='architecture of each discriminator')
        parser.add_argument('--num_D', type=int, default=2, help='number of discriminators to be used in multiscale')
        NLayerDiscriminator.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.cdtype = compute_dtype_of(opt)
        if opt.netD_subarch != 'n_layer':
            raise ValueError('unrecognized discriminator subarchitecture %s' % opt.netD_subarch)
        for i in range(opt.num_D):
            self.add_module('discriminator_%d' % i, NLayerDiscriminator(opt))

    def _side_stream(self, device):
        st = self.__dict__.get('_side')
        if st is None or st.device != device:
            st = self.__dict__['_side'] = torch.cuda.Stream(device=device)
        return st

    def forward(self, input, feat_lambda=None):
        """input: (2N, label_nc+output_nc, H, W) as in the reference call (pix2pix_model.py:338), or
        the already-built (2N,H,W,8) NHWC tensor.  Returns list[num_D] of list[n_layers_D+1] tensors,
        logical NCHW (NHWC storage); list[num_D][1] with --no_ganFeat_loss (discriminator.py:53-63).
        feat_lambda (not in the reference; used by Pix2PixModel's G step): the batch is [fake | real] and the GAN
        feature-matching loss  lambda/num_D * sum_ij L1mean(fake_ij, real_ij.detach())  is computed on the way
        (ops.feat_tap); returns (result, loss[1]) with the intermediate features of `result` detached."""
        self.require_gpu(input)
        x = input if (input.dim() == 4 and input.shape[-1] == D_CPAD and input.shape[1] != self.opt.label_nc + self.opt.output_nc) \
            else to_d_input(input, self.cdtype)
        bank = sn_begin(self)                  # one batched power iteration for both scales' SN convs
        with packing.network_scope(self, bank):    # all weight packs of this forward: one launch
            result = []
            keep_all = not self.opt.no_ganFeat_loss
            terms = [] if feat_lambda is not None else None
            num_D = len(list(self.named_children()))
            # G step ([fake | real], real = detached feature-matching target): gradients exist for the fake half only
            # (not with BatchNorm in D: batch statistics couple the samples, the real half's activations then DO carry gradient)
            coupled = any(isinstance(m, nn.BatchNorm2d) for m in self.modules())
            with ops.LivePrefix.of(x.shape[0] // 2 if (terms is not None and not coupled) else None):
                children = list(self.named_children())
                if _TWO_STREAMS and len(children) == 2 and x.is_cuda:
                    # The two scales are independent chains (reference discriminator.py:53-63 runs them one after the other): the
                    # half-resolution one -- 65^2 ... 18^2 maps, launches of a few dozen workgroups -- runs on a SECOND stream beside
                    # the full-resolution one (its ~30 small launches per pass no longer queue behind each other's
                    # tails).  Autograd runs each node's backward on its forward's stream and joins the streams when a gradient
                    # crosses; inside a hipGraph capture the fork / join become parallel branches of the graph.
                    main, side = torch.cuda.current_stream(), self._side_stream(x.device)
                    x2 = ops.avgpool3x3s2(x)                   # F.avg_pool2d(3, 2, 1, count_include_pad=False)
                    side.wait_stream(main)
                    terms2 = [] if terms is not None else None
                    with torch.cuda.stream(side):
                        raw2 = children[1][1].forward_nhwc(x2, terms2, (feat_lambda or 0.0) / num_D)
                    raw1 = children[0][1].forward_nhwc(x, terms, (feat_lambda or 0.0) / num_D)
                    main.wait_stream(side)
                    x2.record_stream(side)
                    for t in raw2 + (terms2 or []):
                        t.record_stream(main)                  # (allocated on the side stream, consumed by the loss code on this one)
                    if terms is not None:
                        terms += terms2
                    for raw in (raw1, raw2):
                        feats = [f.permute(0, 3, 1, 2) for f in raw]
                        result.append(feats if keep_all else [feats[-1]])
                else:
                    for name, D in children:
                        raw = D.forward_nhwc(x, terms, (feat_lambda or 0.0) / num_D)
                        feats = [f.permute(0, 3, 1, 2) for f in raw]
                        result.append(feats if keep_all else [feats[-1]])
                        x = ops.avgpool3x3s2(x)                # F.avg_pool2d(3, 2, 1, count_include_pad=False)
            if terms is None:
                return result
            return result, torch.stack(terms).sum().view(1)
