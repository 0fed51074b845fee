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
    """construct -> print -> move to the GPU -> init_weights (models/networks/__init__.py:39-48).
    One process drives one GPU, so there is no nn.DataParallel wrap; multi-GPU is
    seg2eye_amd.distributed (RCCL gradient all-reduce)."""
    net = cls(opt)
    net.print_network()
    if len(opt.gpu_ids) > 0:
        if not torch.cuda.is_available():
            raise RuntimeError('opt.gpu_ids=%s but no GPU is visible' % (opt.gpu_ids,))
        net.cuda()
    net.init_weights(opt.init_type, opt.init_variance)
    return net


def define_G(opt):
    return create_network(SPADESTYLEGenerator, opt)


def define_D(opt):
    return create_network(MultiscaleDiscriminator, opt)


def define_E(opt):
    return create_network(ConvEncoder, opt)
