"""SPADE+Style block and the style FC (reference models/networks/normalization.py).

Parameter containers keep the reference's attribute names so state_dict keys match
(`spade.mlp_shared.0.weight`, `spade.mlp_gamma.weight`, `adain.linear.weight`, ...); the forward is
three HIP launches: label-gather conv (mlp_shared on the nearest-downsampled label map), one
implicit-GEMM conv producing [gamma | beta], and the fused modulation kernel."""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class SegMap:
    """The segmentation input of the generator as the kernels want it: a uint8 label map
    (N,H,W) on the GPU.  Built once per forward from whatever the caller passed: the one-hot
    float tensor of the reference API (pix2pix_model.py:146-154), a (N,1,H,W)/(N,H,W) label map,
    or another SegMap."""

    def __init__(self, label):
        assert label.dtype == torch.uint8 and label.dim() == 3
        self.label = label.contiguous()

    @property
    def shape(self):
        return tuple(self.label.shape)

    @staticmethod
    def of(seg):
        if isinstance(seg, SegMap):
            return seg
        if seg.dim() == 4 and seg.shape[1] > 1 and seg.is_floating_point():      # one-hot (N,nc,H,W)
            return SegMap(seg.argmax(dim=1).to(torch.uint8))
        if seg.dim() == 4 and seg.shape[1] == 1:
            seg = seg[:, 0]
        return SegMap(seg.to(torch.uint8))


class SPADE(nn.Module):
    """Parameter holder for SPADE (normalization.py:63-105).  `instance` is the variant the hot path is defined on
    (SURVEY F2); `batch` -- the reference's default `--norm_G spectralspadebatch3x3` -- normalises with the statistics of
    the whole batch and keeps BatchNorm2d's running buffers (same state_dict keys: `param_free_norm.running_mean`,
    `.running_var`, `.num_batches_tracked`, registered first like in the reference).  `syncbatch` raises there too."""

    def __init__(self, config_text, norm_nc, label_nc):
        super().__init__()
        assert config_text.startswith('spade')
        parsed = re.search(r'spade(\D+)(\d)x\d', config_text)
        kind, ks = str(parsed.group(1)), int(parsed.group(2))
        if kind not in ('instance', 'batch'):
            raise ValueError('%s is not a recognized param-free norm type in SPADE' % kind)
        if ks != 3:
            raise ValueError('SPADE kernel size %d not supported (3 only)' % ks)
        self.kind = kind
        if kind == 'batch':
            self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)
        nhidden = 128
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, nhidden, kernel_size=3, padding=1), nn.ReLU())
        self.mlp_gamma = nn.Conv2d(nhidden, norm_nc, kernel_size=3, padding=1)
        self.mlp_beta = nn.Conv2d(nhidden, norm_nc, kernel_size=3, padding=1)

    def gamma_beta(self, seg, h, w, dtype):
        """[gamma | beta] as one (N,h,w,2C) tensor."""
        return ops.spade_params(seg.label, self.mlp_shared[0].weight, self.mlp_shared[0].bias,
                                self.mlp_gamma.weight, self.mlp_gamma.bias, self.mlp_beta.weight, self.mlp_beta.bias,
                                h, w, dtype)

    def arena_order(self):
        """Parameter order that makes [W_gamma; W_beta] and [b_gamma; b_beta] contiguous in a flat arena."""
        return [self.mlp_shared[0].weight, self.mlp_shared[0].bias, self.mlp_gamma.weight, self.mlp_beta.weight,
                self.mlp_gamma.bias, self.mlp_beta.bias]


def spade_stats(x, spades, replication=1):
    """{mean, rstd} (N,C,2) fp32 of x for the SPADE modules that normalise it (norm_0 and norm_s of a block share x).
    replication: the modules will see every pixel of x `replication` times (x is about to go through the nearest 2x
    upsampling: 4) -- mean and biased variance are those of x itself, so the pass runs over a quarter of the data; only
    BatchNorm's unbiased running variance needs the real count.
    instance: per-sample statistics (one pass over x).  batch, train mode: statistics of the whole batch, combined from
    the same pass's per-sample fp64 sums; every module's running buffers are updated like nn.BatchNorm2d does (momentum
    0.1, unbiased running variance).  batch, eval mode: a module's own running statistics -- they differ between
    modules, so this returns None when more than one is asked for (each block then calls again for itself)."""
    sp = spades[0]
    if sp.kind == 'instance':
        return ops.in_stats(x.detach())
    n, h, w, c = x.shape
    if not sp.training:
        if len(spades) > 1:
            return None
        bn = sp.param_free_norm
        st = torch.stack([bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)], -1)
        return st.unsqueeze(0).expand(n, c, 2).contiguous()
    _, sums = ops.in_stats(x.detach(), return_sums=True)
    cnt = float(n * h * w * replication)
    tot = sums.sum(0) * float(replication)                        # (C,2) fp64
    from .. import distributed as sdist
    if sdist.sync_world_size() > 1:
        # data-parallel replicas normalise with the statistics of the GLOBAL batch (the per-layer 2*C exchange a
        # synchronised BatchNorm needs; normalization.py:74-75 sees the whole batch on the reference's single GPU).  Equal
        # shards: the count is world * local.  Running buffers then evolve identically on every replica.
        # (sync_world_size, not world_size: rank 0's validation passes run alone -- distributed.solo -- and use local counts)
        sdist.all_reduce_sum_(tot)
        cnt *= sdist.sync_world_size()
    mean = tot[:, 0] / cnt
    var = (tot[:, 1] / cnt - mean * mean).clamp_min(0.0)
    with torch.no_grad():
        for m in spades:
            bn = m.param_free_norm
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_((var * (cnt / max(cnt - 1.0, 1.0))).to(bn.running_var.dtype), alpha=mom)
            bn.num_batches_tracked += 1
    st = torch.stack([mean, torch.rsqrt(var + sp.param_free_norm.eps)], -1).float()
    return st.unsqueeze(0).expand(n, c, 2).contiguous()


class FC(nn.Module):
    """Dense layer of the style path (normalization.py:108-141): fp32 torch ops on (N, w_dim)."""

    def __init__(self, in_channels, out_channels, gain=2 ** 0.5, use_wscale=False, lrmul=1.0, bias=True):
        super().__init__()
        he_std = gain * in_channels ** (-0.5)
        if use_wscale:
            init_std = 1.0 / lrmul
            self.w_lrmul = he_std * lrmul
        else:
            init_std = he_std / lrmul
            self.w_lrmul = lrmul
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels) * init_std)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
            self.b_lrmul = lrmul
        else:
            self.bias = None

    def forward(self, x):
        w = self.weight if self.w_lrmul == 1.0 else self.weight * self.w_lrmul
        b = None if self.bias is None else (self.bias if self.b_lrmul == 1.0 else self.bias * self.b_lrmul)
        return F.leaky_relu(F.linear(x.float(), w, b), 0.2)


class ApplyStyle(nn.Module):
    """Holds the style FC (normalization.py:144-169); its affine x*(s0+1)+s1 is applied inside the
    fused modulation kernel."""

    def __init__(self, latent_size, channels, use_wscale):
        super().__init__()
        self.linear = FC(latent_size, channels * 2, gain=1.0, use_wscale=use_wscale)


class SPADE_STYLE_Block(nn.Module):
    """out = (SPADE(x, seg) + ApplyStyle(x, w)) / 2  (normalization.py:172-192), optionally with the
    LeakyReLU(0.2) that follows it in the ResBlk (architecture.py:61) fused in.
    x: (N,h,w,C) NHWC compute-dtype tensor.  stats: in_stats(x) if the caller already has them."""

    def __init__(self, fin, opt):
        super().__init__()
        self.spade = SPADE(opt.norm_G.replace('spectral', ''), fin, opt.semantic_nc)
        self.adain = ApplyStyle(opt.w_dim, channels=fin, use_wscale=False)

    def takes_fold(self, x_low):
        """Does the fused launch take this layer with x handed over BEFORE the nearest 2x upsampling (forward(up=True))?"""
        return ops.spade_fused_supported(x_low, self.spade.mlp_shared[0].out_channels, 8)

    def forward(self, x, segmap, latent_style, stats=None, lrelu=False, relay=False, up=False):
        """relay (not in the reference): also return an alias x' of x for the other consumers of x, see
        ops.spade_style_modulate.
        up (not in the reference; stats given): x is the tensor BEFORE the generator's nearest 2x upsampling (generator.py:77-92);
        the fused launch reads it at (y/2, x/2), its backward returns the gradient w.r.t. that tensor: the upsampled tensor and
        its gradient never exist.  Layers the fused launch does not take upsample first."""
        seg = SegMap.of(segmap)
        n, h, w, c = x.shape
        sp = self.spade
        if up:
            if stats is None:
                raise ValueError('SPADE_STYLE_Block: up=True needs the statistics of x (input_stats(x, 4))')
            h, w = 2 * h, 2 * w
        if stats is None:
            stats = spade_stats(x, [sp])
        batch = sp.kind == 'batch'
        from . import stylebank
        sb = stylebank.current()                                    # inside a generator: all style FCs were one GEMM
        banked = sb is not None and id(self.adain.linear) in sb[0]
        if banked:
            style, kw = sb[1], dict(off=sb[0][id(self.adain.linear)], dbig=sb[2])
        else:
            style, kw = self.adain.linear(latent_style), {}         # (N, 2C) fp32
        fl = 8 if up else 0
        if ops.spade_fused_supported(x, sp.mlp_shared[0].out_channels, fl) and not (up and batch):
            # the big layers: [gamma | beta] conv and modulation in ONE launch, gamma / beta never written (ops.SpadeFusedFn)
            return ops.spade_style_fused(x, seg.label, sp.mlp_shared[0].weight, sp.mlp_shared[0].bias, sp.mlp_gamma.weight,
                                         sp.mlp_gamma.bias, sp.mlp_beta.weight, sp.mlp_beta.bias, style, stats, lrelu,
                                         batch=batch, relay=relay, flags=fl, **kw)
        if up:
            x = ops.upsample2x(x)
        gb = sp.gamma_beta(seg, h, w, x.dtype)
        return ops.spade_style_modulate(x, gb, style, stats, lrelu, batch=batch, relay=relay, **kw)


def get_nonspade_norm_layer(opt, norm_type='instance'):
    """normalization.py:15-47: 'spectralinstance' (the hot path), 'spectralbatch', and the same without 'spectral' / with
    'none'.  Returns add_norm_layer(conv) -> nn.Sequential(conv[, IN])
    with the conv's bias removed when a norm follows, so state_dict keys match (`model1.0.0.weight_orig`)."""
    def add_norm_layer(layer):
        sub = norm_type
        if norm_type.startswith('spectral'):
            layer = nn.utils.spectral_norm(layer)
            sub = norm_type[len('spectral'):]
        if sub == 'none' or len(sub) == 0:
            return layer
        if getattr(layer, 'bias', None) is not None:
            delattr(layer, 'bias')
            layer.register_parameter('bias', None)
        if sub == 'instance':
            return nn.Sequential(layer, nn.InstanceNorm2d(layer.out_channels, affine=False))
        if sub == 'batch':                                          # normalization.py:38-39
            return nn.Sequential(layer, nn.BatchNorm2d(layer.out_channels, affine=True))
        raise ValueError('normalization layer %s is not recognized' % sub)
    return add_norm_layer


def apply_nonspade_norm(h, norm, lrelu):
    """The norm layer get_nonspade_norm_layer put behind a conv (+ the LeakyReLU(0.2) that follows it in netD), on an NHWC
    tensor.  InstanceNorm2d -- the hot path's `spectralinstance` -- is the fused HIP statistics + modulation pass;
    BatchNorm2d (`--norm_D / --norm_E spectralbatch`, normalization.py:38-39, off the benchmarked path) runs on stock
    PyTorch-ROCm: nn.BatchNorm2d on the channels-last view, fp32, with its affine parameters and running buffers (per-GPU
    statistics under data parallelism, as nn.DataParallel would give)."""
    if isinstance(norm, nn.BatchNorm2d):
        y = norm(h.permute(0, 3, 1, 2).float())
        if lrelu:
            y = F.leaky_relu(y, 0.2)
        return y.permute(0, 2, 3, 1).contiguous().to(h.dtype)
    return ops.instance_norm(h, lrelu=lrelu)
