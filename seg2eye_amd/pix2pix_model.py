"""Pix2PixModel: mode-dispatched model of the reference (models/pix2pix_model.py:13-374) for the
G+D train step and inference, over the HIP networks.

Same constructor, forward(data, mode) modes and return types, create_optimizers, save,
loss-log helpers and netG/netD/netE attributes.  Deliberate, documented differences:
  * labels stay uint8 on the GPU (no one-hot tensor is materialised for G; D's 5-channel input is built
    by one kernel), and 3-D (N,H,W) labels are handled correctly for N > 1;
  * the per-sample python loop over netE (pix2pix_model.py:280-290) is one batched call, with the
    same number of spectral-norm power iterations (ConvEncoder.forward(power_iterations=N));
  * netE is loaded whenever netG is;
  * D's weight gradients are not computed during the G step (the reference computes and then zeroes
    them, trainers/pix2pix_trainer.py:38) -- the parameters after the step are identical.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import checkpoint, networks, ops
from ._lib import LOSS_L1
from .networks.base_network import compute_dtype_of
from .networks.normalization import SegMap
from .optim import FlatAdam


class Pix2PixModel(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        networks.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.cdtype = compute_dtype_of(opt)
        self.netG, self.netD, self.netE = self.initialize_networks(opt)
        self.netE.logvar_used = False        # (`logvar` enters no loss: pix2pix_model.py:271-305; see ConvEncoder.forward)
        if opt.isTrain:
            self.criterionGAN = networks.GANLoss(opt.gan_mode, opt=opt)
            if not opt.no_vgg_loss:
                raise NotImplementedError('VGGLoss does not exist in the reference either (SURVEY F1); '
                                          'keep --no_vgg_loss')
            self.reset_loss_log()

    # ------------------------------------------------------------------ loss log (pix2pix_model.py:49-59)
    def get_loss_log(self):
        return {k: torch.mean(torch.stack(v)) for k, v in self.loss_log.items() if len(v)}

    def add_to_loss_log(self, key, value):
        self.loss_log.setdefault(key, []).append(value)

    def reset_loss_log(self):
        self.loss_log = {}

    # ------------------------------------------------------------------ entry point
    def forward(self, data, mode):
        seg, style_image, target_image = self.preprocess_input(data)
        if mode == 'generator':
            return self.compute_generator_loss(seg, style_image, target_image)
        elif mode == 'discriminator':
            return self.compute_discriminator_loss(seg, style_image, target_image)
        elif mode == 'encode_only':
            w, _ = self.encode_w(style_image)
            return w
        elif mode == 'inference':
            with torch.no_grad():
                if 'latent_style' in data:
                    fake_image = self.generate_fake_from_stylecode(seg, data['latent_style'].to(self.device()))
                else:
                    fake_image, _, _ = self.generate_fake(seg, style_image)
                self.reset_loss_log()
            return fake_image
        raise ValueError("|mode| is invalid")

    def create_optimizers(self, opt):
        """TTUR: betas (0, 0.9), lr_G = lr/2 over netG+netE, lr_D = 2*lr (pix2pix_model.py:92-110)."""
        # same parameter SET as the reference (netG + netE).  The ORDER inside the flat arena serves three things:
        #  * each SPADE's gamma / beta conv weights (and biases) back to back, so [gamma | beta] is one zero-copy matrix;
        #  * the style FCs of all SPADE+Style layers back to back (weights, then biases): one GEMM for all (networks/stylebank.py);
        #  * the blocks grouped by WHEN their gradients become final in the backward pass -- (conv_img, up_3, up_2) first, then
        #    (up_1, up_0), G_middle_1, G_middle_0, head_0, then everything that completes at the very end (fc, the style FCs,
        #    netE) -- so that a data-parallel run can all-reduce a group's contiguous slice while the backward is still running
        #    (distributed.FlatGradSync.launch; self.grad_groups_G = the element ranges, in completion order;
        #    networks/generator.py reports group i when the gradient w.r.t. the input of its earliest block exists).
        from .networks.normalization import SPADE, SPADE_STYLE_Block
        fcs = [m.adain.linear for m in self.netG.modules() if isinstance(m, SPADE_STYLE_Block)]
        fc_params = [f.weight for f in fcs] + [f.bias for f in fcs if f.bias is not None]
        seen = {id(q) for q in fc_params}

        def block_params(mods):
            out = []
            for top in mods:
                for mod in top.modules():
                    if isinstance(mod, SPADE):
                        for q in mod.arena_order():
                            if id(q) not in seen:
                                seen.add(id(q))
                                out.append(q)
                for q in top.parameters():
                    if id(q) not in seen:
                        seen.add(id(q))
                        out.append(q)
            return out
        G = self.netG
        # completion groups, in the order the backward finishes them (round 4: the three 1024-channel blocks -- 85 % of the bytes --
        # leave one by one instead of riding with everything that completes at the very end)
        stages = [('conv_img', 'up_3', 'up_2'), ('up_1', 'up_0'), ('G_middle_1',), ('G_middle_0',), ('head_0',)]
        staged = [block_params([m for m in (getattr(G, n, None) for n in names) if m is not None]) for names in stages]
        dead = list(self.netE.fc_var.parameters())          # logvar enters no loss: never a gradient (torch's Adam skips them)
        dead_ids = {id(q) for q in dead}
        early = block_params([G]) + fc_params + [q for q in self.netE.parameters() if id(q) not in dead_ids]   # fc, style FCs, netE
        G_params = [q for grp in staged for q in grp] + early
        self._arena_groups_G_dead = dead
        self._arena_groups_G = [len(grp) for grp in staged] + [len(early)]
        if opt.no_TTUR:
            beta1, beta2, G_lr, D_lr = opt.beta1, opt.beta2, opt.lr, opt.lr
        else:
            beta1, beta2, G_lr, D_lr = 0.0, 0.9, opt.lr / 2, opt.lr * 2
        optimizer_G = FlatAdam(G_params, lr=G_lr, betas=(beta1, beta2), weight_decay=opt.weight_decay, never_updated=dead)
        bounds, k = [0], 0
        for cnt in self._arena_groups_G:
            k += cnt
            bounds.append(optimizer_G.offsets[k] if k < len(optimizer_G.offsets) else optimizer_G.numel)
        bounds[-1] = optimizer_G.numel                              # (the never-updated tail rides with the last group)
        self.grad_groups_G = [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
        optimizer_D = FlatAdam(list(self.netD.parameters()), lr=D_lr, betas=(beta1, beta2),
                               weight_decay=opt.weight_decay) if opt.isTrain else None
        return optimizer_G, optimizer_D

    def save(self, epoch):
        checkpoint.save_network(self.netG, 'G', epoch, self.opt)
        checkpoint.save_network(self.netD, 'D', epoch, self.opt)
        checkpoint.save_network(self.netE, 'E', epoch, self.opt)

    # ------------------------------------------------------------------ helpers
    def device(self):
        return next(self.netG.parameters()).device

    def initialize_networks(self, opt):
        netG = networks.define_G(opt)
        netD = networks.define_D(opt) if opt.isTrain else None
        netE = networks.define_E(opt)
        if not opt.isTrain or opt.continue_train:
            checkpoint.load_network(netG, 'G', opt.which_epoch, opt)
            checkpoint.load_network(netE, 'E', opt.which_epoch, opt)
            if opt.isTrain:
                checkpoint.load_network(netD, 'D', opt.which_epoch, opt)
        return netG, netD, netE

    def preprocess_input(self, data):
        """Move to the GPU; the label map becomes a uint8 SegMap instead of a one-hot float tensor
        (pix2pix_model.py:138-160)."""
        dev = self.device()
        label = data['label']
        if label.dim() == 4:
            label = label[:, 0]
        elif label.dim() == 2:
            label = label.unsqueeze(0)
        seg = SegMap(label.to(dev).to(torch.uint8))
        style = data['style_image'].to(dev)
        target = data['target'].to(dev) if 'target' in data else None
        return seg, style, target

    def compute_generator_loss(self, seg, style_image, target_image):
        G_losses = {}
        opt = self.opt
        style_terms = bool(opt.lambda_style_feat or opt.lambda_style_w or opt.lambda_gram)
        fake_image, latent_style_real, style_features_real = self.generate_fake(seg, style_image, aggregate_features=style_terms)
        d_params = list(self.netD.parameters())
        flags = [p.requires_grad for p in d_params]
        for p in d_params:                      # D's weight grads are dead in the G step
            p.requires_grad_(False)
        try:
            fused_feat = not self.opt.no_ganFeat_loss
            pred, feat = self.discriminate(seg, fake_image, target_image, feat_lambda=self.opt.lambda_feat if fused_feat else None,
                                           divide=False)
        finally:
            for p, f in zip(d_params, flags):
                p.requires_grad_(f)
        gan = self.criterionGAN.undivided(pred, for_discriminator=False)      # (the hinge loss straight from the [fake | real] batch)
        G_losses['GAN'] = gan if gan is not None else self.criterionGAN(self.divide_pred(pred)[0], True, for_discriminator=False)
        if opt.lambda_l2:                                    # pix2pix_model.py:196-200 (nn.MSELoss)
            l2 = F.mse_loss(fake_image.float(), target_image.float()).view(1)
            G_losses['L2/weighted'] = l2 * opt.lambda_l2
            self.add_to_loss_log('L2/raw', l2.detach())
        if self.opt.lambda_l1:
            a = fake_image.permute(0, 2, 3, 1)
            b = target_image.to(a.dtype).permute(0, 2, 3, 1).contiguous()
            l1 = ops.loss_sum(a, b, LOSS_L1, 1.0 / a.numel()).view(1)
            G_losses['L1/weighted'] = l1 * self.opt.lambda_l1
            self.add_to_loss_log('L1/raw', l1.detach())
        if getattr(self.opt, 'lambda_openeds', 0):
            # pix2pix_model.py:206-210: the OpenEDS metric of the batch (per image; no gradient -- the reference's
            # `.int()` cuts the graph too), weighted into the logged loss
            from .networks.loss import MSECalculator
            oe = MSECalculator.calculate_mse_for_tensors(fake_image, target_image)
            G_losses['openeds/weighted'] = oe * self.opt.lambda_openeds
            self.add_to_loss_log('openeds/raw', oe.detach())
        if style_terms:
            # Style consistency (pix2pix_model.py:162-184, 212-229): the generated image is encoded again and compared with
            # the encoding of the style images -- latent code (MSE), per-layer feature maps (MSE), Gram matrices (MSE,
            # target detached by StyleLoss).  The reference's other `.detach()` calls there discard their result, so
            # gradients flow through both encodings; mirrored.  Features are aggregated over the style dimension.
            latent_style_fake, style_features_fake = self.encode_w(fake_image.unsqueeze(1), aggregate_features=True)
            if opt.lambda_style_w > 0:
                raw = F.mse_loss(latent_style_fake.float(), latent_style_real.float()).view(1)
                G_losses['style_w/weighted'] = raw * opt.lambda_style_w
                self.add_to_loss_log('style_w/raw', raw.detach())
            if opt.lambda_style_feat > 0:
                raw = torch.stack([F.mse_loss(a.float(), b.float()) for a, b in zip(style_features_fake, style_features_real)]).sum().view(1)
                G_losses['style_feat/weighted'] = raw * opt.lambda_style_feat
                self.add_to_loss_log('style_feat/raw', raw.detach())
            if opt.lambda_gram > 0:
                raw = torch.stack([F.mse_loss(networks.gram_matrix(a.float()), networks.gram_matrix(b.float()).detach())
                                   for a, b in zip(style_features_fake, style_features_real)]).sum().view(1)
                G_losses['gram/weighted'] = raw * opt.lambda_gram
                self.add_to_loss_log('gram/raw', raw.detach())
        if fused_feat:
            # == networks.feature_matching_loss(pred_fake, pred_real, lambda_feat), computed inside netD's forward so
            # that its gradient is accumulated in place into the features' incoming gradients (ops.FeatTapFn)
            G_losses['GAN_Feat'] = feat
        return G_losses, fake_image

    def compute_discriminator_loss(self, seg, real_image, target_image):
        with torch.no_grad():
            fake_image, _, _ = self.generate_fake(seg, real_image)
        fake_image = fake_image.detach()
        pred, _ = self.discriminate(seg, fake_image, target_image, divide=False)
        both = self.criterionGAN.undivided(pred, for_discriminator=True)
        if both is not None:
            return {'D/Fake': both[0], 'D/real': both[1]}
        pred_fake, pred_real = self.divide_pred(pred)
        return {'D/Fake': self.criterionGAN(pred_fake, False, for_discriminator=True),
                'D/real': self.criterionGAN(pred_real, True, for_discriminator=True)}

    def _aggregate(self, t, dim=1):
        if self.opt.style_aggr_method == 'mean':
            return torch.mean(t, dim=dim)
        if self.opt.style_aggr_method == 'max':
            return torch.max(t, dim=dim).values
        raise ValueError('Aggregation method not found: %s' % self.opt.style_aggr_method)

    def encode_w(self, real_image, aggregate_features=False):
        """(N, input_ns, 1, h, w) style images -> w (N, w_dim) = aggregate over the style dimension of
        netE's mu (pix2pix_model.py:271-314), as ONE batched netE call.
        aggregate_features: the second result is the reference's `features_aggregated` (:297-303) as one (N,C,h,w) tensor
        per encoder layer (aggregated over the style dimension); otherwise the raw (N*ns,C,h,w) layer outputs."""
        if real_image.dim() != 5:
            raise ValueError('real_image should have 5 dimensions')
        n, ns = real_image.shape[:2]
        mu, _, feats = self.netE(real_image.reshape(n * ns, *real_image.shape[2:]), power_iterations=n)
        if aggregate_features:
            feats = [self._aggregate(f.unflatten(0, (n, ns)), dim=1) for f in feats]
        return self._aggregate(mu.view(n, ns, -1)), feats

    def generate_fake_from_stylecode(self, seg, latent_style):
        return self.netG(seg, latent_style)

    def generate_fake(self, seg, style_image, aggregate_features=False):
        latent_style, feats = self.encode_w(style_image, aggregate_features)
        return self.generate_fake_from_stylecode(seg, latent_style), latent_style, feats

    def discriminate(self, seg, fake_image, real_image, feat_lambda=None, divide=True):
        """D on cat over the batch of [cat(seg, fake); cat(seg, real)] (pix2pix_model.py:328-342); the
        (2N,H,W,8) input is built by two launches from the label map and the two image batches.
        divide=False (the loss code of this class): -> (undivided predictions, feature-matching term or None)."""
        x = ops.d_input(seg.label, fake_image.to(self.cdtype), real_image, self.opt.label_nc, networks.discriminator.D_CPAD)
        out, feat = self.netD(x, feat_lambda=feat_lambda) if feat_lambda is not None else (self.netD(x), None)
        if not divide:
            return out, feat
        return self.divide_pred(out) if feat_lambda is None else (self.divide_pred(out), feat)

    @staticmethod
    def divide_pred(pred):
        halves = [[ops.split_halves(t) if t.requires_grad else (t[:t.size(0) // 2], t[t.size(0) // 2:]) for t in p] for p in pred]
        return [[h[0] for h in p] for p in halves], [[h[1] for h in p] for p in halves]

    def use_gpu(self):
        return len(self.opt.gpu_ids) > 0
