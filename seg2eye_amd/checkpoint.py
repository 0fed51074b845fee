"""Checkpoint naming and key compatibility of the reference (util/util.py:195-221):
<checkpoints_dir>/<name>/<epoch>_net_<G|D|E>.pth holding net.state_dict() on the CPU; a DataParallel
'module.' prefix on either side is tolerated."""
import os

import torch


def _path(label, epoch, opt):
    return os.path.join(opt.checkpoints_dir, opt.name, '%s_net_%s.pth' % (epoch, label))


def save_network(net, label, epoch, opt):
    path = _path(label, epoch, opt)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, path)
    return path


def load_network(net, label, epoch, opt):
    sd = torch.load(_path(label, epoch, opt), map_location='cpu')
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    with torch.no_grad():
        own = net.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError('checkpoint lacks keys: %s' % missing[:5])
        for k, v in own.items():
            v.copy_(sd[k])          # in place: parameters may alias an optimizer arena
    return net
