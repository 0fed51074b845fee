"""CPU ORACLE for the Seg2Eye G+D train-step hot path.  TEST INFRASTRUCTURE ONLY.

This is a functional restatement (plain functions over a ``{state-dict key:
tensor}`` mapping, NCHW, torch CPU ops) of the reference's algorithm for the
path BASELINE.json's north_star names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it -- as the checker, never as the product.  Nothing under ``seg2eye_amd/``
imports this file.

Parity pin: the reference holds no tests or golden vectors for this path
(SURVEY section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``tests/golden/make_golden.py``
(which imports /root/reference with stub modules) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every function
here against those vectors.

Third-party arithmetic: the reference's arithmetic lives in PyTorch
(requirements.txt:1 ``torch>=1.0.0``, unpinned): conv2d, InstanceNorm2d
(biased variance, eps 1e-5, no affine), F.interpolate nearest / bilinear,
F.avg_pool2d(count_include_pad=False), nn.utils.spectral_norm (1 power
iteration, eps 1e-12) and optim.Adam.  torch is present on both boxes, so the
oracle calls the same primitives where the reference does and restates the
composite ones (spectral norm, Adam, losses) explicitly.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
import math
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers

def spectral_normalize(weight_orig, u, v, training, eps=1e-12):
    """torch/nn/utils/spectral_norm.py ``compute_weight`` (n_power_iterations=1,
    dim=0) as applied at models/networks/architecture.py:30-34 and
    normalization.py:25-26.  Returns (W, u', v'); u', v' are the post-forward
    buffers (unchanged in eval mode).  sigma is differentiable w.r.t.
    weight_orig with u, v constants."""
    wm = weight_orig.reshape(weight_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
            u = u.clone()
            v = v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return weight_orig / sigma, u, v


def _sn_conv_weight(sd, prefix, training, updates):
    w, u, v = spectral_normalize(sd[prefix + '.weight_orig'], sd[prefix + '.weight_u'],
                                 sd[prefix + '.weight_v'], training)
    if training and updates is not None:
        updates[prefix + '.weight_u'] = u
        updates[prefix + '.weight_v'] = v
    return w


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False): biased variance over (H,W) per (n,c)
    (models/networks/normalization.py:73, :41)."""
    return F.instance_norm(x, eps=eps)


def one_hot_labels(label, label_nc):
    """models/pix2pix_model.py:146-154 (scatter_ one-hot) for 4-D labels."""
    label = label.long()
    if label.dim() == 3:
        label = label.unsqueeze(0)          # reference quirk, SURVEY F4
    bs, _, h, w = label.shape
    out = torch.zeros(bs, label_nc, h, w, dtype=torch.float32)
    return out.scatter_(1, label, 1.0)


# ----------------------------------------------------------------------------- generator

def style_fc(sd, prefix, w):
    """FC.forward models/networks/normalization.py:135-141 with w_lrmul =
    b_lrmul = 1 (use_wscale=False, lrmul=1: :118-131)."""
    return F.leaky_relu(F.linear(w, sd[prefix + '.weight'], sd[prefix + '.bias']), 0.2)


def apply_style(sd, prefix, x, w):
    """ApplyStyle.forward normalization.py:163-169: x*(s0+1)+s1, x NOT normalised."""
    style = style_fc(sd, prefix + '.linear', w).view(-1, 2, x.shape[1], 1, 1)
    return x * (style[:, 0] + 1.0) + style[:, 1]


def spade(sd, prefix, x, seg, training=True, updates=None):
    """SPADE.forward normalization.py:91-105.  param_free_norm is InstanceNorm2d (no state) or, when the state dict
    carries `param_free_norm.running_mean`, BatchNorm2d(affine=False) (normalization.py:72-75): batch statistics and
    a running-buffer update in train mode, the running buffers in eval mode."""
    rm_key = prefix + '.param_free_norm.running_mean'
    if rm_key in sd:
        rm, rv = sd[rm_key].clone(), sd[prefix + '.param_free_norm.running_var'].clone()
        normalized = F.batch_norm(x, rm, rv, None, None, training, 0.1, 1e-5)
        if training and updates is not None:
            updates[rm_key] = rm
            updates[prefix + '.param_free_norm.running_var'] = rv
            updates[prefix + '.param_free_norm.num_batches_tracked'] = sd[prefix + '.param_free_norm.num_batches_tracked'] + 1
    else:
        normalized = instance_norm(x)
    segmap = F.interpolate(seg, size=x.shape[2:], mode='nearest')
    actv = F.relu(F.conv2d(segmap, sd[prefix + '.mlp_shared.0.weight'],
                           sd[prefix + '.mlp_shared.0.bias'], padding=1))
    gamma = F.conv2d(actv, sd[prefix + '.mlp_gamma.weight'], sd[prefix + '.mlp_gamma.bias'], padding=1)
    beta = F.conv2d(actv, sd[prefix + '.mlp_beta.weight'], sd[prefix + '.mlp_beta.bias'], padding=1)
    return normalized * (1 + gamma) + beta


def spade_style_block(sd, prefix, x, seg, w, training=True, updates=None):
    """SPADE_STYLE_Block.forward normalization.py:184-192."""
    return (spade(sd, prefix + '.spade', x, seg, training, updates) + apply_style(sd, prefix + '.adain', x, w)) / 2


def spade_style_resblk(sd, prefix, x, seg, w, training, updates):
    """SPADE_STYLE_ResnetBlock.forward/shortcut architecture.py:44-62; the
    learned shortcut exists iff fin != fout (:19,26-27)."""
    learned = (prefix + '.conv_s.weight_orig') in sd
    if learned:
        ws = _sn_conv_weight(sd, prefix + '.conv_s', training, updates)
        x_s = F.conv2d(spade_style_block(sd, prefix + '.norm_s', x, seg, w, training, updates), ws)
    else:
        x_s = x
    w0 = _sn_conv_weight(sd, prefix + '.conv_0', training, updates)
    dx = F.conv2d(F.leaky_relu(spade_style_block(sd, prefix + '.norm_0', x, seg, w, training, updates), 0.2),
                  w0, sd[prefix + '.conv_0.bias'], padding=1)
    w1 = _sn_conv_weight(sd, prefix + '.conv_1', training, updates)
    dx = F.conv2d(F.leaky_relu(spade_style_block(sd, prefix + '.norm_1', dx, seg, w, training, updates), 0.2),
                  w1, sd[prefix + '.conv_1.bias'], padding=1)
    return x_s + dx


def generator_forward(sd, seg, w, sh, sw, training=False, updates=None, more=False):
    """SPADESTYLEGenerator.forward models/networks/generator.py:69-102
    ('normal'; 'more' adds one upsample between the middle blocks)."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')     # nn.Upsample(scale_factor=2) :50
    x = F.interpolate(seg, size=(sh, sw))                               # nearest, :72
    x = F.conv2d(x, sd['fc.weight'], sd['fc.bias'], padding=1)
    x = spade_style_resblk(sd, 'head_0', x, seg, w, training, updates)
    x = up(x)
    x = spade_style_resblk(sd, 'G_middle_0', x, seg, w, training, updates)
    if more:
        x = up(x)
    x = spade_style_resblk(sd, 'G_middle_1', x, seg, w, training, updates)
    for name in ('up_0', 'up_1', 'up_2', 'up_3'):
        x = up(x)
        x = spade_style_resblk(sd, name, x, seg, w, training, updates)
    x = F.conv2d(F.leaky_relu(x, 0.2), sd['conv_img.weight'], sd['conv_img.bias'], padding=1)
    return torch.tanh(x)


# ----------------------------------------------------------------------------- discriminator

def nlayer_discriminator_forward(sd, prefix, x, n_layers, training, updates):
    """NLayerDiscriminator.forward models/networks/discriminator.py:74-116:
    model0 conv4x4 s2 p2 + lrelu; model1..n-1 SN conv4x4 (s2, last s1, bias
    removed normalization.py:31-35) + IN + lrelu; model_n conv4x4 s1 p2."""
    feats = []
    h = F.leaky_relu(F.conv2d(x, sd[prefix + '.model0.0.weight'], sd[prefix + '.model0.0.bias'],
                              stride=2, padding=2), 0.2)
    feats.append(h)
    for n in range(1, n_layers):
        stride = 1 if n == n_layers - 1 else 2
        wn = _sn_conv_weight(sd, '%s.model%d.0.0' % (prefix, n), training, updates)
        h = F.leaky_relu(instance_norm(F.conv2d(h, wn, None, stride=stride, padding=2)), 0.2)
        feats.append(h)
    k = '%s.model%d.0' % (prefix, n_layers)
    h = F.conv2d(h, sd[k + '.weight'], sd[k + '.bias'], stride=1, padding=2)
    feats.append(h)
    return feats


def discriminator_forward(sd, x, num_D=2, n_layers=4, training=False, updates=None,
                          intermediate=True):
    """MultiscaleDiscriminator.forward/downsample discriminator.py:46-63."""
    result = []
    for i in range(num_D):
        out = nlayer_discriminator_forward(sd, 'discriminator_%d' % i, x, n_layers, training, updates)
        result.append(out if intermediate else [out[-1]])
        x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    return result


# ----------------------------------------------------------------------------- encoder

def encoder_forward(sd, x, training=False, updates=None):
    """ConvEncoder.forward models/networks/encoder.py:53-73: bilinear to 256^2,
    n x (SN conv3x3 s2 p1 no-bias + IN) with NO activation in between, lrelu,
    fc_mu / fc_var."""
    if x.shape[2] != 256 or x.shape[3] != 256:
        x = F.interpolate(x, size=(256, 256), mode='bilinear')
    n_layers = 0
    while ('layer%d.0.weight_orig' % n_layers) in sd:
        n_layers += 1
    feats = []
    h = x
    for i in range(n_layers):
        wi = _sn_conv_weight(sd, 'layer%d.0' % i, training, updates)
        h = instance_norm(F.conv2d(h, wi, None, stride=2, padding=1))
        feats.append(h)
    out = F.leaky_relu(h, 0.2).reshape(h.shape[0], -1)
    mu = F.linear(out, sd['fc_mu.weight'], sd['fc_mu.bias'])
    logvar = F.linear(out, sd['fc_var.weight'], sd['fc_var.bias'])
    return mu, logvar, feats


def _aggregate(t, aggr, dim):
    """_aggregate_tensor models/pix2pix_model.py:271-278."""
    if aggr == 'mean':
        return t.mean(dim=dim)
    if aggr == 'max':
        return t.max(dim=dim).values
    raise ValueError('Aggregation method not found: %s' % aggr)


def encode_w(sdE, style_image, aggr='mean', training=False, updates=None, with_features=False):
    """_compute_multiple_netE / _compute_aggregated_w / encode_w
    models/pix2pix_model.py:280-314: netE is called once PER SAMPLE (a python
    loop; each call is one spectral-norm power iteration in train mode), mu is
    aggregated over the style dimension.  with_features: also the per-sample
    list of per-layer feature maps, each aggregated over the style dimension
    (pix2pix_model.py:297-303)."""
    mus, feats_all = [], []
    cur = dict(sdE)
    for b in range(style_image.shape[0]):
        upd = {} if training else None
        mu, _, feats = encoder_forward(cur, style_image[b], training, upd)
        if training:
            cur.update(upd)
            if updates is not None:
                updates.update(upd)
        mus.append(mu)
        feats_all.append([_aggregate(f, aggr, 0) for f in feats])
    multiple_w = torch.stack(mus, dim=0)                    # (bs, ns, w_dim)
    w = _aggregate(multiple_w, aggr, 1)
    return (w, feats_all) if with_features else w


def gram_matrix(x):
    """models/networks/loss.py:177-189."""
    a, b, c, d = x.size()
    f = x.reshape(a * b, c * d)
    return torch.mm(f, f.t()).div(a * b * c * d)


def style_consistency_losses(w_fake, feats_fake, w_real, feats_real, opt):
    """The three optional terms of compute_generator_loss that re-encode the generated image
    (models/pix2pix_model.py:162-184, 212-229; StyleLoss loss.py:192-199).  Note the reference's
    `.detach()` calls there discard their result: gradients flow through BOTH encodings, except for the
    gram target, which StyleLoss detaches."""
    out = {}
    n_maps = len(feats_fake[0])
    stack = lambda feats, i: torch.stack([feats[b][i] for b in range(len(feats))])
    if opt.lambda_style_w > 0:
        out['style_w/weighted'] = F.mse_loss(w_fake, w_real) * opt.lambda_style_w
    if opt.lambda_style_feat > 0:
        raw = torch.sum(torch.stack([F.mse_loss(stack(feats_fake, i), stack(feats_real, i)) for i in range(n_maps)]))
        out['style_feat/weighted'] = raw * opt.lambda_style_feat
    if opt.lambda_gram > 0:
        raw = torch.sum(torch.stack([F.mse_loss(gram_matrix(stack(feats_fake, i)), gram_matrix(stack(feats_real, i)).detach())
                                     for i in range(n_maps)]))
        out['gram/weighted'] = raw * opt.lambda_gram
    return out


# ----------------------------------------------------------------------------- losses

def hinge_loss(pred, target_is_real, for_discriminator=True):
    """GANLoss.loss hinge branch models/networks/loss.py:66-77."""
    if for_discriminator:
        if target_is_real:
            return -torch.mean(torch.min(pred - 1, torch.zeros_like(pred)))
        return -torch.mean(torch.min(-pred - 1, torch.zeros_like(pred)))
    assert target_is_real
    return -torch.mean(pred)


def gan_loss(preds, target_is_real, for_discriminator=True):
    """GANLoss.__call__ loss.py:85-99 on the list-of-lists: last output of each
    scale, averaged over scales; result has shape [1]."""
    loss = 0
    for pred_i in preds:
        t = hinge_loss(pred_i[-1], target_is_real, for_discriminator)
        loss = loss + t.view(1, -1).mean(dim=1)
    return loss / len(preds)


def divide_pred(pred):
    """Pix2PixModel.divide_pred models/pix2pix_model.py:345-358."""
    fake = [[t[:t.shape[0] // 2] for t in p] for p in pred]
    real = [[t[t.shape[0] // 2:] for t in p] for p in pred]
    return fake, real


def feature_matching_loss(pred_fake, pred_real, lambda_feat=10.0):
    """GAN_Feat in compute_generator_loss models/pix2pix_model.py:231-241."""
    num_D = len(pred_fake)
    total = torch.zeros(1)
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            total = total + F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat / num_D
    return total


# ----------------------------------------------------------------------------- model-level

# ---------------------------------------------------------------------------------------------- OpenEDS metric
def to_255(image):
    """data/postprocessor.py:58-73 (`ImageProcessor.unnormalize`, the branch taken for images in [-1, 1]):
    add 1, multiply by 255, divide by 2 (fp32, in this order), then `.int()` -- truncation toward zero."""
    return torch.div(torch.mul(torch.add(image.float(), 1), 255), 2).int()


def openeds_accuracy(produced, target):
    """models/networks/loss.py:102-111: sqrt(sum((p - t)^2)) / (h * w) of ONE image (any leading dims are summed too)."""
    diff = produced.float() - target.float()
    h, w = diff.shape[-2:]
    return torch.sqrt(torch.sum(diff * diff).float()) / (h * w)


def mse_for_images(produced, target):
    """loss.py:116-133 (`MSECalculator.calculate_mse_for_images`): per-image error of 0..255 images (N,C,H,W) -> (N,)."""
    return torch.stack([openeds_accuracy(produced[i], target[i]) for i in range(produced.shape[0])])


def mse_for_tensors(produced, target):
    """loss.py:135-155 (`calculate_mse_for_tensors`, = criterionOpenEDS, pix2pix_model.py:36): both in [-1, 1], mapped to
    0..255 ints first; the result carries no gradient (`.int()`)."""
    return mse_for_images(to_255(produced), to_255(target))


def error_statistics(all_errors, mode, dataset_key):
    """loss.py:157-171: {'mse/<key>/<mode>/relative': sum(errors) / len(errors) * 1471}."""
    import numpy as np
    all_errors = np.asarray(all_errors, dtype=np.float64)
    return {'mse/%s/%s/relative' % (dataset_key, mode): float(np.sum(all_errors) / len(all_errors) * 1471)}


def _cv2_linear_taps(n_dst, n_src):
    """The tap table of OpenCV's resize for INTER_LINEAR (modules/imgproc/src/resize.cpp, `cv::resize` -> the generic
    `resizeGeneric_` path): for destination index d,
        scale = 1 / (n_dst / n_src)                      (double)
        f  = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s              (float)
        s < 0            -> f = 0, s = 0                                          (left / top edge clamp)
        s >= n_src - 1   -> f = 0, s = n_src - 1                                  (right / bottom edge clamp)
        weights (1 - f, f) as FLOAT (`AT = float` also for CV_64F images); the second tap is s + 1 clamped into the image
    -> (s0, s1, w0, w1): int64 indices, float32 weights."""
    import numpy as np
    inv = float(n_dst) / float(n_src)
    scale = 1.0 / inv
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    s1 = np.minimum(s + 1, n_src - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize_bilinear(image, w=400, h=640):
    """data/postprocessor.py:108-114 (`ImageProcessor.resize`): per image, `cv2.resize(img.astype(np.float), (w, h),
    interpolation=cv2.INTER_LINEAR)` -- a float64 image through OpenCV's bilinear resize.
    THIRD-PARTY, ABSENT HERE: opencv-python (requirements.txt: `opencv-python`, unpinned) is not in this image and cannot be
    installed, so this function cannot be checked against cv2 itself: it is PARITY-UNPINNED against the library and pinned
    only against its published algorithm, restated explicitly below (no torch / scipy resampling call in between):
    half-pixel centres, edge clamp, float32 tap weights (`_cv2_linear_taps`), a horizontal pass then a vertical pass, both
    in float64 (`WT = double` for CV_64F), `S0 * w0 + S1 * w1` with the weights promoted to double.  No antialiasing when
    shrinking.  -> float64 tensor (N, C, h, w)."""
    import numpy as np
    x = image.detach().cpu().double().numpy()
    H, W = x.shape[-2:]
    xs0, xs1, xa0, xa1 = _cv2_linear_taps(w, W)
    ys0, ys1, yb0, yb1 = _cv2_linear_taps(h, H)
    rows = x[..., :, xs0] * xa0.astype(np.float64) + x[..., :, xs1] * xa1.astype(np.float64)            # horizontal pass
    out = rows[..., ys0, :] * yb0.astype(np.float64)[:, None] + rows[..., ys1, :] * yb1.astype(np.float64)[:, None]
    return torch.from_numpy(out)


def to_255_pre_truncation(image, w=400, h=640):
    """The float64 value `unnormalize` truncates (postprocessor.py:58-73 on the resized float64 image): ((x + 1) * 255) / 2."""
    return torch.div(torch.mul(torch.add(resize_bilinear(image, w, h), 1), 255), 2)


def to_255_resized(image, w=400, h=640):
    """postprocessor.py:92-97 (`to_255resized_imagebatch`): resize, then unnormalize."""
    return to_255_pre_truncation(image, w, h).int()


class OracleModel:
    """Functional counterpart of Pix2PixModel + Pix2PixTrainer
    (models/pix2pix_model.py, trainers/pix2pix_trainer.py) over three state
    dicts.  Always 'train mode' like the reference unless
    ``training=False`` is passed explicitly."""

    def __init__(self, sdG, sdD, sdE, opt, sh, sw):
        self.G = {k: v.clone() for k, v in sdG.items()}
        self.D = {k: v.clone() for k, v in sdD.items()} if sdD is not None else None
        self.E = {k: v.clone() for k, v in sdE.items()}
        self.opt, self.sh, self.sw = opt, sh, sw
        self.adam = {}        # name -> (m, v)
        self.steps = {'G': 0, 'D': 0}

    # -- parameter bookkeeping -------------------------------------------------
    @staticmethod
    def is_param(key):
        leaf = key.rsplit('.', 1)[-1]
        return leaf in ('weight', 'bias', 'weight_orig')

    def _leaf(self, sd):
        out = {}
        for k, v in sd.items():
            out[k] = v.clone().requires_grad_(True) if self.is_param(k) else v
        return out

    # -- forward pieces (models/pix2pix_model.py:316-342) ----------------------
    def generate_fake(self, G, E, seg, style, training, updG, updE, with_features=False):
        enc = encode_w(E, style, self.opt.style_aggr_method, training, updE, with_features)
        w, feats = enc if with_features else (enc, None)
        fake = generator_forward(G, seg, w, self.sh, self.sw, training, updG,
                                 more=(self.opt.num_upsampling_layers == 'more'))
        return (fake, w, feats) if with_features else (fake, w)

    def discriminate(self, D, seg, fake, real, training, updD):
        fake_and_real = torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0)
        out = discriminator_forward(D, fake_and_real, self.opt.num_D, self.opt.n_layers_D,
                                    training, updD, intermediate=not self.opt.no_ganFeat_loss)
        return divide_pred(out)

    def generator_losses(self, G, D, E, data, training=True, updates=None):
        """compute_generator_loss models/pix2pix_model.py:186-247 (default
        lambdas: GAN + GAN_Feat)."""
        updG, updD, updE = ({}, {}, {}) if updates is None else updates
        seg = one_hot_labels(data['label'], self.opt.label_nc)
        opt = self.opt
        style_terms = bool(getattr(opt, 'lambda_style_feat', 0) or getattr(opt, 'lambda_style_w', 0) or getattr(opt, 'lambda_gram', 0))
        fake, w_real, feats_real = self.generate_fake(G, E, seg, data['style_image'], training, updG, updE, True)
        pred_fake, pred_real = self.discriminate(D, seg, fake, data['target'], training, updD)
        losses = {'GAN': gan_loss(pred_fake, True, for_discriminator=False)}
        if getattr(opt, 'lambda_l2', 0):                                   # pix2pix_model.py:196-200
            losses['L2/weighted'] = F.mse_loss(fake, data['target']) * opt.lambda_l2
        if getattr(opt, 'lambda_l1', 0):                                   # :201-205
            losses['L1/weighted'] = F.l1_loss(fake, data['target']) * opt.lambda_l1
        if getattr(opt, 'lambda_openeds', 0):                              # :206-210 (per image; carries no gradient)
            losses['openeds/weighted'] = mse_for_tensors(fake.detach(), data['target']) * opt.lambda_openeds
        if style_terms:                                                    # :212-229: second encode, u/v keep iterating
            E2 = {**E, **updE} if training else E
            w_fake, feats_fake = encode_w(E2, fake.unsqueeze(1), opt.style_aggr_method, training, updE, True)
            losses.update(style_consistency_losses(w_fake, feats_fake, w_real, feats_real, opt))
        if not self.opt.no_ganFeat_loss:
            losses['GAN_Feat'] = feature_matching_loss(pred_fake, pred_real, self.opt.lambda_feat)
        return losses, fake

    def discriminator_losses(self, G, D, E, data, training=True, updates=None):
        """compute_discriminator_loss models/pix2pix_model.py:249-264."""
        updG, updD, updE = ({}, {}, {}) if updates is None else updates
        seg = one_hot_labels(data['label'], self.opt.label_nc)
        with torch.no_grad():
            fake, _ = self.generate_fake(G, E, seg, data['style_image'], training, updG, updE)
        fake = fake.detach()
        pred_fake, pred_real = self.discriminate(D, seg, fake, data['target'], training, updD)
        return {'D/Fake': gan_loss(pred_fake, False, True), 'D/real': gan_loss(pred_real, True, True)}

    # -- optimizer (torch.optim.Adam semantics; pix2pix_model.py:92-110) -------
    def _adam(self, tag, named_params, grads, lr, beta1, beta2, eps=1e-8):
        self.steps[tag] += 1
        t = self.steps[tag]
        for name, p in named_params:
            g = grads[name]
            if g is None:
                continue
            key = tag + ':' + name
            m, v = self.adam.get(key, (torch.zeros_like(p), torch.zeros_like(p)))
            m = beta1 * m + (1 - beta1) * g
            v = beta2 * v + (1 - beta2) * g * g
            self.adam[key] = (m, v)
            bc1 = 1 - beta1 ** t
            bc2 = 1 - beta2 ** t
            denom = v.sqrt() / math.sqrt(bc2) + eps
            p.data.addcdiv_(m, denom, value=-(lr / bc1))

    def _hyper(self):
        if self.opt.no_TTUR:
            return self.opt.beta1, self.opt.beta2, self.opt.lr, self.opt.lr
        return 0.0, 0.9, self.opt.lr / 2, self.opt.lr * 2

    def run_generator_one_step(self, data):
        """Pix2PixTrainer.run_generator_one_step trainers/pix2pix_trainer.py:26-35."""
        G, D, E = self._leaf(self.G), self._leaf(self.D), self._leaf(self.E)
        upd = ({}, {}, {})
        losses, fake = self.generator_losses(G, D, E, data, True, upd)
        total = sum(losses.values()).mean()
        names = [('G.' + k, G[k]) for k in G if self.is_param(k)] + \
                [('E.' + k, E[k]) for k in E if self.is_param(k)]
        grads = torch.autograd.grad(total, [p for _, p in names], allow_unused=True)
        b1, b2, lrG, _ = self._hyper()
        self._adam('G', names, dict(zip([n for n, _ in names], grads)), lrG, b1, b2)
        for sd, leaf, u in ((self.G, G, upd[0]), (self.D, D, upd[1]), (self.E, E, upd[2])):
            for k in sd:
                if self.is_param(k):
                    sd[k] = leaf[k].detach()
            for k, v in u.items():
                sd[k] = v.detach()
        return {k: v.detach() for k, v in losses.items()}, fake.detach()

    def run_discriminator_one_step(self, data):
        """Pix2PixTrainer.run_discriminator_one_step trainers/pix2pix_trainer.py:37-45."""
        G, D, E = dict(self.G), self._leaf(self.D), dict(self.E)
        upd = ({}, {}, {})
        losses = self.discriminator_losses(G, D, E, data, True, upd)
        total = sum(losses.values()).mean()
        names = [('D.' + k, D[k]) for k in D if self.is_param(k)]
        grads = torch.autograd.grad(total, [p for _, p in names], allow_unused=True)
        b1, b2, _, lrD = self._hyper()
        self._adam('D', names, dict(zip([n for n, _ in names], grads)), lrD, b1, b2)
        for k in self.D:
            if self.is_param(k):
                self.D[k] = D[k].detach()
        for sd, u in ((self.G, upd[0]), (self.D, upd[1]), (self.E, upd[2])):
            for k, v in u.items():
                sd[k] = v.detach()
        return {k: v.detach() for k, v in losses.items()}
