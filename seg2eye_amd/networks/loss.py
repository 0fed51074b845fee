"""GAN losses (reference models/networks/loss.py:17-99 and the feature-matching term of
pix2pix_model.py:231-241) as HIP reductions."""
import torch
import torch.nn as nn

from .. import ops
from .._lib import LOSS_NEG_MEAN, LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_L1


def _flat(t):
    """Any tensor whose storage order is irrelevant for an element-wise reduction -> contiguous."""
    if t.is_contiguous():
        return t
    p = t.permute(0, 2, 3, 1) if t.dim() == 4 else t
    return p if p.is_contiguous() else t.contiguous()


class GANLoss(nn.Module):
    """hinge GAN loss with the reference's list-of-lists handling (loss.py:85-99): the last output of
    every scale, averaged over scales; returns a tensor of shape [1]."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=None, opt=None):
        super().__init__()
        if gan_mode not in ('hinge', 'ls', 'original', 'w'):
            raise ValueError('Unexpected gan_mode {}'.format(gan_mode))
        self.gan_mode = gan_mode
        self.real_label, self.fake_label = target_real_label, target_fake_label
        self.opt = opt

    def loss(self, input, target_is_real, for_discriminator=True, weight=1.0, pooled=False):
        """weight (not in the reference): a factor folded into the reduction's scale -- the 1 / num_D of the list form below,
        instead of a tensor division after it."""
        if self.gan_mode != 'hinge':
            return weight * self._loss_other(input, target_is_real) if weight != 1.0 else self._loss_other(input, target_is_real)
        x = _flat(input)
        n = x.numel()
        if for_discriminator:
            mode = LOSS_HINGE_REAL if target_is_real else LOSS_HINGE_FAKE
        else:
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            mode = LOSS_NEG_MEAN
        return ops.loss_sum(x, None, mode, weight / n, pooled)

    def undivided(self, pred, for_discriminator):
        """The same losses from the UNDIVIDED predictions of a [fake | real] batch (list over scales of lists whose last entry
        is the prediction): for_discriminator -> (loss on the fake half aiming for fake, loss on the real half aiming for
        real) = __call__(pred_fake, False), __call__(pred_real, True); else the generator's loss on the fake half =
        __call__(pred_fake, True, for_discriminator=False).  None when this mode / layout has no undivided form (the caller
        then divides, as the reference does, pix2pix_model.py:344-358)."""
        if self.gan_mode != 'hinge' or not isinstance(pred, list):
            return None
        last = [p[-1] if isinstance(p, list) else p for p in pred]
        flats = [_flat(t) for t in last]
        if any((not f.is_contiguous()) or f.shape[0] % 2 for f in flats):
            return None
        k = len(flats)
        totals = None
        for f in flats:
            w = 1.0 / k / (f.numel() // 2)
            if for_discriminator:
                terms = ops.pair_loss(f, LOSS_HINGE_FAKE, LOSS_HINGE_REAL, w, pooled=k > 1)
            else:
                terms = (ops.half_loss(f, False, LOSS_NEG_MEAN, w, pooled=k > 1),)
            terms = [t.view(1) for t in terms]
            totals = terms if totals is None else [a + b for a, b in zip(totals, terms)]
        return tuple(totals) if for_discriminator else totals[0]

    def _loss_other(self, input, target_is_real):
        # the non-default modes (loss.py:58-65, 78-83) act on the few-thousand-element PatchGAN outputs: plain torch
        x = input.float()
        if self.gan_mode == 'w':
            return -x.mean() if target_is_real else x.mean()
        target = torch.full_like(x, self.real_label if target_is_real else self.fake_label)
        if self.gan_mode == 'original':
            return torch.nn.functional.binary_cross_entropy_with_logits(x, target)
        return torch.nn.functional.mse_loss(x, target)

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            # mean over the scales of each scale's loss (loss.py:85-99), the 1 / num_D folded into each reduction's scale: the
            # list form costs one add per extra scale instead of `0 + l`, adds and a division (each a launch, each with a backward)
            total = None
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                # (a term that is summed with others right here may accumulate in the step's zero pool: ops.loss_sum)
                term = self.loss(pred_i, target_is_real, for_discriminator, weight=1.0 / len(input), pooled=len(input) > 1).view(1)
                total = term if total is None else total + term
            return total
        return self.loss(input, target_is_real, for_discriminator)


def feature_matching_loss(pred_fake, pred_real, lambda_feat):
    """sum_i sum_{j<last} L1mean(fake_ij, real_ij.detach()) * lambda_feat / num_D, shape [1]."""
    num_D = len(pred_fake)
    total = None
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            a, b = _flat(pred_fake[i][j]), _flat(pred_real[i][j].detach())
            term = ops.loss_sum(a, b, LOSS_L1, lambda_feat / num_D / a.numel())
            total = term if total is None else total + term
    return total.view(1)


def gram_matrix(x):
    """(N,C,h,w) -> (N*C, N*C) Gram matrix / (N*C*h*w)  (reference models/networks/loss.py:177-189)."""
    a, b, c, d = x.size()
    f = x.reshape(a * b, c * d)
    return torch.mm(f, f.t()).div(a * b * c * d)


def openEDSaccuracy(produced, target):
    """models/networks/loss.py:102-111 for ONE image pair already in 0..255 (any numeric dtype, on the GPU):
    sqrt(sum (p - t)^2) / (h * w)."""
    from .. import ops
    p8 = produced.reshape(1, *produced.shape[-2:]).to(torch.uint8)
    t8 = target.reshape(1, *target.shape[-2:]).to(torch.uint8)
    return ops.openeds_error_u8(p8, t8)[0]


class MSECalculator:
    """The reference's OpenEDS metric (models/networks/loss.py:114-171), computed on the device by `s2e_openeds_error*`
    instead of a Python loop over CPU images.  Same method names, argument meaning and checks."""

    @classmethod
    def calculate_mse_for_images(cls, produced, target):
        assert produced.shape == target.shape
        assert produced.shape[-2:] == (640, 400), 'Invalid shape: %s' % (tuple(produced.shape),)
        assert len(produced.shape) == 4, 'Please feed 4D tensors'
        if produced.dtype != torch.uint8:
            assert float(produced.min()) >= 0 and float(produced.max()) <= 255
            assert float(target.min()) >= 0 and float(target.max()) <= 255
        from .. import ops
        return ops.openeds_error_u8(produced.to(torch.uint8), target.to(torch.uint8))

    @classmethod
    def calculate_mse_for_tensors(cls, produced, target):
        assert produced.shape == target.shape
        assert len(produced.shape) == 4, 'Please feed 4D tensors'
        from .. import ops
        return ops.openeds_error(produced, target)

    @classmethod
    def calculate_error_statistics(cls, all_errors, mode, dataset_key):
        import numpy as np
        all_errors = np.asarray(all_errors, dtype=np.float64)
        return {'mse/%s/%s/relative' % (dataset_key, mode): float(np.sum(all_errors) / len(all_errors) * 1471)}
