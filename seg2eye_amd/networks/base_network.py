"""BaseNetwork: the class surface define_G/define_D/define_E rely on
(reference models/networks/base_network.py:10-59): modify_commandline_options,
print_network, init_weights(init_type, gain)."""
import torch
import torch.nn as nn
from torch.nn import init

from .. import _lib


def compute_dtype_of(opt):
    name = getattr(opt, 'compute_dtype', 'bf16')
    if name in ('bf16', 'bfloat16'):
        return torch.bfloat16
    if name in ('fp32', 'float32'):
        return torch.float32
    raise ValueError('compute_dtype must be bf16 or fp32, got %r' % (name,))


def _fill(w, init_type, gain):
    if init_type == 'normal':
        init.normal_(w, 0.0, gain)
    elif init_type == 'xavier':
        init.xavier_normal_(w, gain=gain)
    elif init_type == 'xavier_uniform':
        init.xavier_uniform_(w, gain=1.0)
    elif init_type == 'kaiming':
        init.kaiming_normal_(w, a=0, mode='fan_in')
    elif init_type == 'orthogonal':
        init.orthogonal_(w, gain=gain)
    else:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)


class BaseNetwork(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print('Network [%s] was created. Total number of parameters: %.1f million. '
              'To see the architecture, do print(network).' % (type(self).__name__, n / 1e6))

    def init_weights(self, init_type='normal', gain=0.02):
        """Conv2d / Linear weights by init_type, their biases to 0; the style FC is not touched
        (its class name matches neither 'Conv' nor 'Linear' in the reference).  For a
        spectral-normed conv the tensor written is weight_orig (the reference writes it through
        the pre-first-forward alias m.weight.data)."""
        for m in self.modules():
            if not isinstance(m, (nn.Conv2d, nn.Linear)):
                continue
            w = getattr(m, 'weight_orig', None)
            if w is None:
                w = m.weight
            with torch.no_grad():
                if init_type == 'none':
                    m.reset_parameters()
                else:
                    _fill(w, init_type, gain)
                if getattr(m, 'bias', None) is not None:
                    m.bias.zero_()

    def require_gpu(self, *tensors):
        for t in tensors:
            if torch.is_tensor(t) and not t.is_cuda:
                raise _lib.Seg2EyeHipError(
                    '%s runs on MI355X HIP kernels only; got a %s tensor (no CPU fallback)' % (type(self).__name__, t.device))
