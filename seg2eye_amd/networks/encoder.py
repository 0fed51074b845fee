"""Style encoder (reference models/networks/encoder.py:13-73): 5-6 x (SN conv3x3 s2 + InstanceNorm)
with NO activation in between, LeakyReLU, two FCs."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import packing
from ..spectral import sn_begin
from .base_network import BaseNetwork, compute_dtype_of
from .normalization import apply_nonspade_norm, get_nonspade_norm_layer


class ConvEncoder(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.cdtype = compute_dtype_of(opt)
        kw = 3
        pw = int(np.ceil((kw - 1.0) / 2))
        ndf = opt.ngf
        norm_layer = get_nonspade_norm_layer(opt, opt.norm_E)
        chans = [1, ndf, ndf * 2, ndf * 4, ndf * 8, ndf * 8]
        if opt.crop_size >= 256:
            chans.append(ndf * 8)
        self.len_sequence = len(chans) - 1
        for n in range(self.len_sequence):
            self.add_module('layer' + str(n), norm_layer(nn.Conv2d(chans[n], chans[n + 1], kw, stride=2, padding=pw)))
        self.so = s0 = 4
        self.fc_mu = nn.Linear(ndf * 8 * s0 * s0, opt.w_dim)
        self.fc_var = nn.Linear(ndf * 8 * s0 * s0, opt.w_dim)
        self.actvn = nn.LeakyReLU(0.2, False)

    def forward(self, x, get_intermediate_features=False, power_iterations=1):
        """x: (M,1,h,w) style images.  Returns (mu, logvar, features) like encoder.py:53-73; features are
        logical-NCHW views of the NHWC layer outputs.
        power_iterations: spectral-norm iterations for this call.  The reference calls netE once per
        SAMPLE (pix2pix_model.py:280-290), i.e. N power iterations per encode; the batched call made by
        Pix2PixModel passes N so u, v follow the same trajectory.  Layer outputs are unaffected beyond
        eps effects: InstanceNorm follows each conv and removes the 1/sigma scale."""
        self.require_gpu(x)
        bank = sn_begin(self, power_iterations)
        with packing.network_scope(self, bank):    # all weight packs of this forward: one launch
            if x.size(2) != 256 or x.size(3) != 256:
                x = F.interpolate(x.float(), size=(256, 256), mode='bilinear')
            h = x.permute(0, 2, 3, 1).contiguous().to(self.cdtype)          # (M,256,256,1): same memory order as NCHW
            feats = []
            for i in range(self.len_sequence):
                blk = getattr(self, 'layer%d' % i)
                conv = blk[0] if isinstance(blk, nn.Sequential) else blk
                h = ops.conv2d_m(h, conv, None, 2, 1)
                if isinstance(blk, nn.Sequential):
                    h = apply_nonspade_norm(h, blk[1], lrelu=False)
                feats.append(h.permute(0, 3, 1, 2))
            out = F.leaky_relu(feats[-1].float(), 0.2).reshape(h.shape[0], -1)     # NCHW flatten order, encoder.py:68
            mu = self.fc_mu(out)
            logvar = self.fc_var(out)
            return mu, logvar, feats
