"""The plugin boundary of the reference (models/networks/__init__.py:39-60): define_G / define_D /
define_E return nn.Modules whose forward signatures and state_dict keys match the reference's, with
the compute in libseg2eye_hip.so."""
import torch
import torch.nn as nn

from .base_network import BaseNetwork
from .discriminator import MultiscaleDiscriminator, NLayerDiscriminator
from .encoder import ConvEncoder
from .generator import SPADESTYLEGenerator
from .loss import GANLoss, feature_matching_loss, gram_matrix
from .normalization import SegMap


def modify_commandline_options(parser, is_train):
    SPADESTYLEGenerator.modify_commandline_options(parser, is_train)
    if is_train:
        MultiscaleDiscriminator.modify_commandline_options(parser, is_train)
    ConvEncoder.modify_commandline_options(parser, is_train)
    return parser


def create_network(cls, opt):
    """The reference's construction sequence (models/networks/__init__.py:39-48): build, print the parameter count, move to
    the GPU named by opt.gpu_ids, initialise the weights.  One process drives one GPU, so there is no nn.DataParallel
    wrap; multi-GPU is seg2eye_amd.distributed (RCCL gradient all-reduce)."""
    on_gpu = len(opt.gpu_ids) > 0
    if on_gpu and not torch.cuda.is_available():
        raise RuntimeError('opt.gpu_ids=%s but no GPU is visible' % (opt.gpu_ids,))
    net = cls(opt)
    net.print_network()
    if on_gpu:
        net.cuda()
    net.init_weights(opt.init_type, opt.init_variance)
    return net


# the three factories the reference's Pix2PixModel calls (models/networks/__init__.py:51-60)
NETWORK_FAMILIES = {'G': SPADESTYLEGenerator, 'D': MultiscaleDiscriminator, 'E': ConvEncoder}


def _factory(kind):
    def define(opt):
        return create_network(NETWORK_FAMILIES[kind], opt)
    define.__name__ = define.__qualname__ = 'define_' + kind
    define.__doc__ = 'opt -> %s on the HIP kernels' % NETWORK_FAMILIES[kind].__name__
    return define


define_G, define_D, define_E = _factory('G'), _factory('D'), _factory('E')
