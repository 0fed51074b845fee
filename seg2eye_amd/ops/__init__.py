"""PyTorch-ROCm custom ops over the C-ABI kernels (include/seg2eye_hip.h).

torch is plumbing here: it owns device memory (caching allocator), the stream and
the autograd tape; every forward/backward body below is one or more HIP kernel
launches through ctypes on torch's current stream.  Internal activations are
NHWC-contiguous 4-D tensors (N, H, W, C) in the compute dtype (bf16 or fp32).

There is no CPU path: calling an op with a non-CUDA tensor raises.

One module per family (round 5; `ops.<name>` keeps resolving for every name, private helpers included):
  core      helpers, LaunchProfiler, ZeroPool / GradSink (the trainer step's scratch and deferred weight-side launches)
  conv      packs, conv2d / wgrad launchers, Conv2dFn, the encoder's FC head
  spade     statistics, label convs, the SPADE+Style block (fused, label-sparse), InstanceNorm
  resample  upsampling, resize, pooling, the discriminator's input plumbing
  losses    loss reductions, feature matching, Adam, the OpenEDS metric
  switches  the experiment switches (environment)"""
from . import switches                                   # noqa: F401
from . import core, conv, spade, resample, losses        # noqa: F401
from .._lib import (NORM_SPADE_STYLE_BATCH, NORM_ACCUMULATE_DX, ConvDesc, ACT_NONE, ACT_LRELU, ACT_TANH, AUX_NONE, AUX_RELU_MASK,      # noqa: F401
                    AUX_LRELU_GRAD, NORM_SPADE_STYLE, NORM_PLAIN_IN, LOSS_NEG_MEAN, LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_L1)
from .. import _lib as L                                 # noqa: F401

for _m in (core, conv, spade, resample, losses):
    for _k, _v in vars(_m).items():
        if not _k.startswith('__') and _k not in ('switches', 'core', 'conv', 'spade', 'resample', 'losses'):
            globals().setdefault(_k, _v)
del _m, _k, _v
