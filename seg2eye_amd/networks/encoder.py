"""Style encoder (reference models/networks/encoder.py:13-73): 5-6 x (SN conv3x3 s2 + InstanceNorm)
with NO activation in between, LeakyReLU, two FCs."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import ACT_LRELU
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
        # Pix2PixModel never uses `logvar` (pix2pix_model.py:271-305) and says so (False): fc_var is then not evaluated (forward
        # returns logvar = None) and its gradient stays None as in the reference
        self.logvar_used = True

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
                h = ops.bilinear_resize(x, 256, 256, self.cdtype)                    # F.interpolate(..., 'bilinear'), encoder.py:54-55
            else:
                h = x.permute(0, 2, 3, 1).contiguous().to(self.cdtype)              # (M,256,256,1): same memory order as NCHW
            feats = []
            for i in range(self.len_sequence):
                blk = getattr(self, 'layer%d' % i)
                conv = blk[0] if isinstance(blk, nn.Sequential) else blk
                h = ops.conv2d_m(h, conv, None, 2, 1)
                if isinstance(blk, nn.Sequential):
                    h = apply_nonspade_norm(h, blk[1], lrelu=False)
                feats.append(h.permute(0, 3, 1, 2))
        # fc_mu / fc_var on leaky_relu(x).view(M, -1) (encoder.py:68-71) as ONE 4x4 valid convolution: flattening (M,C,4,4) in
        # NCHW order and multiplying by a (16, C*16) matrix is a conv with that matrix viewed (16, C, 4, 4); both heads are its 32
        # output channels, the LeakyReLU is the conv's fused input activation.  fp32 (the style code feeds every modulation:
        # the few kFLOP are not worth bf16's three digits).  Outside the pack plan's scope: the concatenated weight is a
        # temporary, packed on the spot.
        # logvar_used False (set by Pix2PixModel, which never uses logvar): in the reference an unused logvar leaves fc_var's
        # gradient None and torch's Adam skips the parameter (through a concatenation it would receive explicit zeros instead,
        # which any optimizer with weight decay would act on: ADVICE r2) ...
        if not self.logvar_used:
            # ... and then fc_var is not evaluated at all: mu alone is fc_mu's weight VIEWED as the conv weight (no concatenation,
            # half the FC work); logvar comes back as None
            mu = ops.fc_head(h, self.fc_mu.weight, self.fc_mu.bias, 0.2)      # one launch each way (csrc/style_fc.hip)
            if mu is None:                                                    # (more than 64 style images / 32 outputs: the conv form)
                wmu = self.fc_mu.weight.view(self.opt.w_dim, h.shape[-1], self.so, self.so)
                mu = ops.conv2d(h.float(), wmu, self.fc_mu.bias, None, 1, 0, ACT_LRELU).reshape(h.shape[0], self.opt.w_dim)
            return mu, None, feats
        wcat = torch.cat([self.fc_mu.weight, self.fc_var.weight], 0).view(2 * self.opt.w_dim, h.shape[-1], self.so, self.so)
        bcat = torch.cat([self.fc_mu.bias, self.fc_var.bias], 0)
        out = ops.conv2d(h.float(), wcat, bcat, None, 1, 0, ACT_LRELU).reshape(h.shape[0], 2 * self.opt.w_dim)
        return out[:, :self.opt.w_dim], out[:, self.opt.w_dim:], feats
