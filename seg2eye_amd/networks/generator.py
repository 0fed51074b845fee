"""SPADE+Style generator (reference models/networks/generator.py:13-101)."""
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import ACT_LRELU, ACT_TANH
from ..options import latent_size
from .. import packing
from . import stylebank
from ..spectral import sn_begin
from .architecture import SPADE_STYLE_ResnetBlock
from .base_network import BaseNetwork, compute_dtype_of
from .normalization import SegMap


_FOLD_UP = os.environ.get('S2E_FOLD_UPSAMPLE', '1') != '0'      # A/B switch: 0 = materialise the upsampled tensor in the no-grad forward too


class SPADESTYLEGenerator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument('--num_upsampling_layers', choices=('normal', 'more'), default='normal',
                            help="If 'more', adds upsampling layer between the two middle resnet blocks "
                                 "('most' is broken in the reference and not offered)")
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.cdtype = compute_dtype_of(opt)
        nf = opt.ngf
        self.sw, self.sh = latent_size(opt)
        self.fc = nn.Conv2d(opt.semantic_nc, 16 * nf, 3, padding=1)
        self.head_0 = SPADE_STYLE_ResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_0 = SPADE_STYLE_ResnetBlock(16 * nf, 16 * nf, opt)
        self.G_middle_1 = SPADE_STYLE_ResnetBlock(16 * nf, 16 * nf, opt)
        self.up_0 = SPADE_STYLE_ResnetBlock(16 * nf, 8 * nf, opt)
        self.up_1 = SPADE_STYLE_ResnetBlock(8 * nf, 4 * nf, opt)
        self.up_2 = SPADE_STYLE_ResnetBlock(4 * nf, 2 * nf, opt)
        self.up_3 = SPADE_STYLE_ResnetBlock(2 * nf, 1 * nf, opt)
        self.conv_img = nn.Conv2d(nf, opt.output_nc, 3, padding=1)
        self.up = nn.Upsample(scale_factor=2)          # kept for attribute parity; forward uses ops.upsample2x

    def forward(self, input, w=None):
        """input: one-hot segmap (N,label_nc,H,W) as in the reference call site
        (pix2pix_model.py:316-318), or a label map / SegMap.  w: (N,w_dim) style code.
        Returns (N,output_nc,H,W) in [-1,1] (compute dtype; NHWC storage)."""
        seg = SegMap.of(input)
        self.require_gpu(seg.label, w)
        n, H, W = seg.shape
        f = 32 if self.opt.num_upsampling_layers == 'normal' else 64
        if (H, W) != (self.sh * f, self.sw * f):
            raise ValueError('label map is %dx%d but this generator emits %dx%d (SURVEY F5)' % (H, W, self.sh * f, self.sw * f))
        w = w.float()
        bank = sn_begin(self)          # one batched power iteration for all 18 spectral-normed convs
        pre = self.__dict__.get('_spade_prepass')
        if pre is None:
            pre = self.__dict__['_spade_prepass'] = ops.SpadePrepass()
        # all weight packs: one launch; all style FCs: one GEMM; all label convs / per-class tables: one launch each
        with packing.network_scope(self, bank), stylebank.scope(self, w), pre.scope(seg.label, self.cdtype):
            # F.interpolate(seg, (sh, sw)) + fc conv, generator.py:72-73
            x = ops.label_conv3x3(seg.label, self.fc.weight, self.fc.bias, self.sh, self.sw, False, self.cdtype)
            # a block that follows an upsampling takes its input statistics from the tensor BEFORE it (nearest 2x
            # replication changes neither mean nor variance): a quarter of the bytes for the same numbers
            # data parallel: when the gradient w.r.t. the input of a stage's EARLIEST block exists, every parameter gradient of
            # that stage is final -- tell the trainer, which starts the stage's all-reduce (and, replaying hipGraphs, closes a
            # graph segment there) while the rest of the backward runs.  Stages = Pix2PixModel.create_optimizers' arena groups:
            # 0 (conv_img, up_3, up_2), 1 (up_1, up_0), 2 G_middle_1, 3 G_middle_0, 4 head_0; the rest leaves after the backward.
            cb = self.__dict__.get('grad_ready') if torch.is_grad_enabled() else None

            def mark(t, i):
                if cb is not None and t.requires_grad:
                    t.register_hook(lambda g, i=i: cb(i))
            mark(x, 4)
            x = self.head_0(x, seg, w)
            st = self.G_middle_0.input_stats(x, 4)
            mark(x, 3)
            x = ops.upsample2x(x)
            x = self.G_middle_0(x, seg, w, st)
            st = None
            mark(x, 2)
            if self.opt.num_upsampling_layers == 'more':
                st = self.G_middle_1.input_stats(x, 4)
                x = ops.upsample2x(x)
            x = self.G_middle_1(x, seg, w, st)
            for blk in (self.up_0, self.up_1, self.up_2, self.up_3):
                st = blk.input_stats(x, 4)
                if blk is self.up_2 or blk is self.up_0:
                    mark(x, 0 if blk is self.up_2 else 1)
                # the nearest 2x upsampling is folded into the block (its two SPADE launches read x at (y/2, x/2), forward and
                # backward: SPADE_STYLE_ResnetBlock.forward(up=True)); a block that cannot fold upsamples first
                x = blk(x, seg, w, st, up=True) if _FOLD_UP else blk(ops.upsample2x(x), seg, w, st)
            # conv_img(leaky_relu(x)) + tanh, generator.py:99-100: one launch
            y = ops.conv2d(x, self.conv_img.weight, self.conv_img.bias, None, 1, 1, ACT_LRELU, ACT_TANH)
            return y.permute(0, 3, 1, 2)
