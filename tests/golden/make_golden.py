#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference).

Runs only in the build container (the reference never travels to the GPU box).
The reference is imported with stub modules for the packages this image lacks
(cv2, h5py, torchvision, scipy.misc -- SURVEY App. B); none of its source is
copied.  Inputs and weights come from seg2eye_amd.synthetic (pure integer-hash
functions), so a fixture stores only: the state-dict manifest (names + shapes),
the small inputs, and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import argparse
import os
import sys
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

import numpy as np
import torch


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    cv2 = mod('cv2', INTER_NEAREST=0, INTER_LINEAR=1, INTER_CUBIC=2, FONT_HERSHEY_SIMPLEX=0)
    cv2.cv2 = cv2
    mod('h5py')

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    tv = mod('torchvision')
    tv.transforms = mod('torchvision.transforms', Normalize=_Dummy, Lambda=_Dummy, Compose=_Dummy,
                        ToTensor=_Dummy, Resize=_Dummy)
    tv.utils = mod('torchvision.utils', make_grid=lambda *a, **k: None)
    import scipy
    scipy.misc = mod('scipy.misc')


def ref_opt(**kw):
    from seg2eye_amd.options import default_opt
    kw.setdefault('gpu_ids', [])
    opt = default_opt(**kw)
    delattr(opt, 'compute_dtype')
    delattr(opt, 'hip_graphs')
    delattr(opt, 'no_overlap_allreduce')
    return opt


def load_filled(net, seed=0):
    from seg2eye_amd.synthetic import fill_state_dict
    sd = net.state_dict()
    manifest = [(k, tuple(v.shape)) for k, v in sd.items()]
    filled = fill_state_dict(manifest, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    return manifest


def manifest_arrays(prefix, manifest):
    return {prefix + '_names': np.array([m[0] for m in manifest]),
            prefix + '_shapes': np.array([list(m[1]) + [0] * (4 - len(m[1])) for m in manifest], np.int64),
            prefix + '_ndims': np.array([len(m[1]) for m in manifest], np.int64)}


def checksum(t):
    t = t.detach().double().flatten()
    # (fp32 linspace as the first fixtures were written with; past 2^24 elements its end point rounds out of range)
    idx = torch.linspace(0, t.numel() - 1, 16, dtype=torch.float32 if t.numel() <= (1 << 24) else torch.float64).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item()], t[idx].numpy()])


def max_aggr_fixture(args, syn):
    """T4: --style_aggr_method max (pix2pix_model.py:271-278; the aggregation the paper's figure states): w = max over the four
    style images of netE's mu, so netE's gradient reaches only the arg-max image per (sample, component).  One G step + one D
    step of the reference trainer, ngf=ndf=8, 256x256, N=2; the style codes themselves are stored too (encode_only)."""
    small_trainer_fixture(args, syn, 'trainer_max_ngf8_256.npz', seed=29, with_w=True, grads=('E',), style_aggr_method='max')


def style_trainer_fixture(args, syn):
    small_trainer_fixture(args, syn, 'trainer_style_ngf8_256.npz', seed=23, lambda_l2=15.0, lambda_l1=2.0, lambda_style_w=0.5,
                          lambda_style_feat=0.001, lambda_gram=10000.0)


def small_trainer_fixture(args, syn, fname, seed, with_w=False, grads=(), **opt_kw):
    """T2: the reference's own training recipe switches the optional losses on (scripts/current_runs_spadestyle.sh,
    run name '..._l2_15_lambda_w_0.5_lambda_feat_0.001_lambda_gram_10000_...'): L2 + the three style-consistency terms
    that re-encode the generated image (pix2pix_model.py:196-229).  One G step + one D step, ngf=ndf=8, 256x256, N=2."""
    class FloatAdam(torch.optim.Adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatAdam
    from trainers.pix2pix_trainer import Pix2PixTrainer
    opt = ref_opt(ngf=8, ndf=8, crop_size=256, aspect_ratio=1.0, batchSize=2, init_type='normal', **opt_kw)
    opt.checkpoints_dir = '/tmp/s2e_golden_ckpt'
    trainer = Pix2PixTrainer(opt)
    model = trainer.pix2pix_model
    mG, mD, mE = load_filled(model.netG), load_filled(model.netD), load_filled(model.netE)
    batch = syn.make_batch(2, 256, 256, seed=seed)

    def tdata():
        return {'label': torch.from_numpy(batch['label'].astype(np.int64)),
                'style_image': torch.from_numpy(batch['style_image']),
                'target': torch.from_numpy(batch['target']), 'filename': batch['filename']}
    rec = {}
    if with_w:
        # the aggregated style codes BEFORE any step (train mode: each netE call advances its power iteration, as in the steps);
        # u, v are put back so that the steps below start from the filled state
        sdE = {k: v.clone() for k, v in model.netE.state_dict().items()}
        with torch.no_grad():
            rec['w'] = model(tdata(), mode='encode_only').numpy()
        model.netE.load_state_dict(sdE)
    trainer.run_generator_one_step(tdata())
    for k, v in trainer.g_losses.items():
        rec['it0_%s' % k.replace('/', '_')] = v.detach().numpy().reshape(-1)
    rec['it0_fake_sub'] = trainer.generated.detach()[:, :, ::8, ::8].numpy()
    for tag in grads:
        for k, q in getattr(model, 'net' + tag).named_parameters():
            if q.grad is not None:
                rec['it0_grad_%s.%s' % (tag, k)] = checksum(q.grad)
    trainer.run_discriminator_one_step(tdata())
    for k, v in trainer.d_losses.items():
        rec['it0_%s' % k.replace('/', '_')] = v.detach().numpy().reshape(-1)
    for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
        for k, v in net.state_dict().items():
            rec['it0_ck_%s.%s' % (tag, k)] = checksum(v)
    np.savez_compressed(os.path.join(args.out, fname), **rec,
                        **manifest_arrays('G', mG), **manifest_arrays('D', mD), **manifest_arrays('E', mE))
    print(fname, 'ok', {k: float(v.reshape(-1)[0]) for k, v in rec.items() if k.startswith('it0_') and v.size == 1})


def bn_generator_fixture(args, syn, onehot, SPADESTYLEGenerator):
    """G3: the reference's DEFAULT --norm_G spectralspadebatch3x3 (BatchNorm SPADE): train-mode forward (batch statistics,
    running buffers updated), every parameter gradient, then an eval-mode forward on the updated running buffers."""
    opt = ref_opt(ngf=8, crop_size=64, aspect_ratio=1.0, norm_G='spectralspadebatch3x3')
    netG = SPADESTYLEGenerator(opt)
    man = load_filled(netG)
    sd0 = {k: v.clone() for k, v in netG.state_dict().items()}
    label = syn.ellipse_labels(3, 64, 64, seed=31)
    w = syn.hash_normal('latent_w', (3, 16), seed=31)
    proj = torch.from_numpy(syn.hash_uniform('g_proj', (3, 1, 64, 64), seed=9))
    netG.train()
    wt = torch.from_numpy(w).clone().requires_grad_(True)
    y = netG(onehot(label), wt)
    (y * proj).sum().backward()
    rec = {'label': label, 'w': w, 'y_train': y.detach().numpy(), 'grad_w': wt.grad.numpy()}
    for k, p in netG.named_parameters():
        rec['grad_' + k] = checksum(p.grad)
    for k, v in netG.state_dict().items():
        if k.endswith(('weight_u', 'weight_v')) or 'param_free_norm' in k:
            rec['buf_' + k] = v.detach().numpy().copy()
    netG.eval()
    with torch.no_grad():
        rec['y_eval_after'] = netG(onehot(label), torch.from_numpy(w)).numpy()
    np.savez_compressed(os.path.join(args.out, 'g_bn_ngf8_64.npz'), **rec, **manifest_arrays('G', man))
    print('bn generator ok', float(y.std()), [k for k in rec if 'running_mean' in k][:1])


def openeds_fixture(args, syn):
    """openeds_metric.npz: the reference's OpenEDS error metric (models/networks/loss.py:102-171 `openEDSaccuracy`,
    `MSECalculator`; data/postprocessor.py:58-97 `ImageProcessor.unnormalize / to_255imagebatch`) on synthetic images.
    (`ImageProcessor.resize` needs cv2, which this image lacks: not part of the fixture.)"""
    from models.networks.loss import MSECalculator, openEDSaccuracy
    from data.postprocessor import ImageProcessor
    a = torch.from_numpy(syn.make_batch(3, 96, 80, seed=501)['target'])            # (3,1,96,80) in [-1,1]
    b = torch.from_numpy(syn.make_batch(3, 96, 80, seed=502)['target'])
    # 0..255 images at the metric's mandatory 640 x 400: synthetic targets through the reference's own unnormalize
    ia = ImageProcessor.to_255imagebatch(torch.from_numpy(syn.make_batch(2, 640, 400, seed=503)['target']))
    ib = ImageProcessor.to_255imagebatch(torch.from_numpy(syn.make_batch(2, 640, 400, seed=504)['target']))
    out = {                                         # inputs are regenerated from the seeds by the tests (seg2eye_amd.synthetic)
        'seeds': np.array([501, 502, 503, 504]),
        'to255_a': ImageProcessor.to_255imagebatch(a).numpy().astype(np.uint8),
        'ia_checksum': np.array([int(ia.long().sum()), int((ia.long() * ia.long()).sum())]),
        'mse_tensors': MSECalculator.calculate_mse_for_tensors(a, b).numpy(),
        'mse_images': MSECalculator.calculate_mse_for_images(ia, ib).numpy(),
        'acc_single': openEDSaccuracy(ia[0], ib[0]).numpy(),
    }
    errs = [0.0123, 0.0456, 0.0101, 0.0320]
    st = MSECalculator.calculate_error_statistics(np.array(errs), mode='full', dataset_key='validation')
    out['stat_errors'] = np.array(errs)
    out['stat_key'] = np.array(list(st.keys())[0])
    out['stat_value'] = np.array(list(st.values())[0])
    np.savez_compressed(os.path.join(args.out, 'openeds_metric.npz'), **out)
    print('openeds_metric.npz', out['mse_tensors'], out['mse_images'], st)


def cfg5_train_fixture(args, syn):
    """T5 (config 5's per-GPU workload): one G step + one D step of the reference trainer at ngf=ndf=64, 640x384
    (--crop_size 384 --aspect_ratio 0.6), N=4, encoder + feature matching on (they always are) -- what cfg3's fixture is for
    256x256.  A few minutes of CPU."""
    full_train_fixture(args, syn, fname='trainer_ngf64_640x384_n4.npz', crop=384, aspect=0.6, n=4, hw=(640, 384), seed=77)


def full_train_fixture(args, syn, fname='trainer_ngf64_256_n8.npz', crop=256, aspect=1.0, n=8, hw=(256, 256), seed=1234):
    """T3 (config 3 AS BENCHED): one G step + one D step of the reference `Pix2PixTrainer`
    (trainers/pix2pix_trainer.py:26-45) at ngf=ndf=64, 256x256, N=8 -- the batch bench.py times (seed 1234).
    Stores the losses, a strided subsample of the generated image, a checksum of every ResBlk output of the G step's
    generator forward, and checksums of every parameter / buffer after the iteration.  A few minutes of CPU."""
    class FloatAdam(torch.optim.Adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatAdam
    from trainers.pix2pix_trainer import Pix2PixTrainer
    import time
    opt = ref_opt(ngf=64, ndf=64, crop_size=crop, aspect_ratio=aspect, batchSize=n, init_type='normal')
    opt.checkpoints_dir = '/tmp/s2e_golden_ckpt'
    trainer = Pix2PixTrainer(opt)
    model = trainer.pix2pix_model
    mG, mD, mE = load_filled(model.netG), load_filled(model.netD), load_filled(model.netE)
    batch = syn.make_batch(n, hw[0], hw[1], seed=seed)

    def tdata():
        return {'label': torch.from_numpy(batch['label'].astype(np.int64)),
                'style_image': torch.from_numpy(batch['style_image']),
                'target': torch.from_numpy(batch['target']), 'filename': batch['filename']}
    rec, hooks = {}, []
    for name in ('head_0', 'G_middle_0', 'G_middle_1', 'up_0', 'up_1', 'up_2', 'up_3'):
        def hook(mod, inp, out, name=name):
            rec.setdefault('it0_act_' + name, checksum(out))           # first call = the G step's forward
        hooks.append(getattr(model.netG, name).register_forward_hook(hook))
    t0 = time.time()
    trainer.run_generator_one_step(tdata())
    for h in hooks:
        h.remove()
    for k, v in trainer.g_losses.items():
        rec['it0_%s' % k.replace('/', '_')] = v.detach().numpy().reshape(-1)
    rec['it0_fake_sub'] = trainer.generated.detach()[:, :, ::8, ::8].numpy()
    for tag, net in (('G', model.netG), ('E', model.netE)):
        for k, p in net.named_parameters():
            if p.grad is not None:
                rec['it0_grad_%s.%s' % (tag, k)] = checksum(p.grad)
    t1 = time.time()
    trainer.run_discriminator_one_step(tdata())
    for k, v in trainer.d_losses.items():
        rec['it0_%s' % k.replace('/', '_')] = v.detach().numpy().reshape(-1)
    for k, p in model.netD.named_parameters():
        if p.grad is not None:
            rec['it0_grad_D.%s' % k] = checksum(p.grad)
    for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
        for k, v in net.state_dict().items():
            rec['it0_ck_%s.%s' % (tag, k)] = checksum(v)
    rec['seconds'] = np.array([t1 - t0, time.time() - t1, torch.get_num_threads()])
    np.savez_compressed(os.path.join(args.out, fname), **rec,
                        **manifest_arrays('G', mG), **manifest_arrays('D', mD), **manifest_arrays('E', mE))
    print(fname, 'ok (G step %.0f s, D step %.0f s)' % (t1 - t0, time.time() - t1),
          {k: float(v.reshape(-1)[0]) for k, v in rec.items() if k.startswith('it0_') and v.size == 1})


def cfg5_fixture(args, syn, onehot, SPADESTYLEGenerator, MultiscaleDiscriminator):
    """C5 (config 5 geometry at full width): the reference netG eval-mode forward at ngf=64, 640x384
    (--crop_size 384 --aspect_ratio 0.6; generator.py:52-67), N=4, and netD's forward lists (ndf=64) on
    [cat(seg, fake); cat(seg, real)] (pix2pix_model.py:328-342)."""
    opt = ref_opt(ngf=64, ndf=64, crop_size=384, aspect_ratio=0.6)
    netG = SPADESTYLEGenerator(opt)
    manG = load_filled(netG)
    H, W = netG.sh * 32, netG.sw * 32
    assert (H, W) == (640, 384), (H, W)
    batch = syn.make_batch(4, H, W, seed=55)
    label = batch['label']
    w = syn.hash_normal('latent_w', (4, 16), seed=55)
    seg = onehot(label)
    netG.eval()
    with torch.no_grad():
        y = netG(seg, torch.from_numpy(w))
    netD = MultiscaleDiscriminator(opt)
    manD = load_filled(netD)
    netD.eval()
    real = torch.from_numpy(batch['target'])
    with torch.no_grad():
        pred = netD(torch.cat([torch.cat([seg, y], 1), torch.cat([seg, real], 1)], 0))
    rec = {'w': w, 'y_sub': y[:, :, ::8, ::8].numpy().astype(np.float32),
           'stats': np.array([y.mean().item(), y.std().item(), y.norm().item()])}
    for i in range(2):
        for j in range(5):
            rec['pred_%d_%d' % (i, j)] = pred[i][j].numpy() if j == 4 else checksum(pred[i][j])
            rec['pred_shape_%d_%d' % (i, j)] = np.array(pred[i][j].shape)
    np.savez_compressed(os.path.join(args.out, 'cfg5_ngf64_640x384_n4.npz'), **rec,
                        **manifest_arrays('G', manG), **manifest_arrays('D', manD))
    print('cfg5 ok: y mean/std', y.mean().item(), y.std().item())


def more_fixture(args, syn, onehot, SPADESTYLEGenerator):
    """--num_upsampling_layers more (generator.py:52-67,80-82: six upsamplings, crop/64 start): eval forward + grads."""
    opt = ref_opt(ngf=8, crop_size=128, aspect_ratio=1.0, num_upsampling_layers='more')
    netG = SPADESTYLEGenerator(opt)
    man = load_filled(netG)
    H, W = netG.sh * 64, netG.sw * 64
    label = syn.ellipse_labels(2, H, W, seed=41)
    w = syn.hash_normal('latent_w', (2, 16), seed=41)
    netG.eval()
    wt = torch.from_numpy(w).clone().requires_grad_(True)
    y = netG(onehot(label), wt)
    proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(y.shape), seed=41))
    (y * proj).sum().backward()
    rec = {'label': label, 'w': w, 'y_eval': y.detach().numpy(), 'grad_w': wt.grad.numpy(), 'hw': np.array([H, W, netG.sh, netG.sw])}
    for k, p in netG.named_parameters():
        rec['grad_' + k] = checksum(p.grad)
    np.savez_compressed(os.path.join(args.out, 'g_more_ngf8_128.npz'), **rec, **manifest_arrays('G', man))
    print('more ok', (H, W), float(y.std()))


def resize_fixture(args, syn):
    """resize_cv2_rule.npz: the oracle's explicit float64 restatement of cv2.INTER_LINEAR + the reference's unnormalize
    (oracle.resize_bilinear / to_255_pre_truncation; cv2 itself is not installable here -- parity-unpinned against the
    library) on two synthetic images: the float64 pre-truncation values of a strided subsample, the full uint8
    results' checksums, and the reference's OWN `unnormalize` (.int() of a float64 tensor) applied to those values, which
    pins the truncation step to the real reference code."""
    sys.path.insert(0, REPO)
    from oracle import seg2eye_oracle as O
    from data.postprocessor import ImageProcessor
    out = {'seeds': np.array([601, 602])}
    for i, (seed, hw) in enumerate(((601, (256, 256)), (602, (640, 384)))):
        img = torch.from_numpy(syn.make_batch(1, hw[0], hw[1], seed=seed)['target'])          # (1,1,H,W) in [-1,1]
        pre = O.to_255_pre_truncation(img)                                                       # float64 (1,1,640,400)
        q_ref = ImageProcessor.unnormalize(O.resize_bilinear(img), as_tensor=True)              # the reference's own truncation
        assert torch.equal(q_ref, pre.int())
        out['pre_sub_%d' % i] = pre[0, 0, ::7, ::5].numpy()
        out['q_sub_%d' % i] = q_ref[0, 0, ::3, ::3].numpy().astype(np.uint8)
        out['q_sums_%d' % i] = np.array([int(q_ref.long().sum()), int((q_ref.long() ** 2).sum())])
        out['hw_%d' % i] = np.array(hw)
    np.savez_compressed(os.path.join(args.out, 'resize_cv2_rule.npz'), **out)
    print('resize ok', {k: v.shape for k, v in out.items()})


def batchnorm_de_fixture(args, syn, onehot, MultiscaleDiscriminator, ConvEncoder):
    """de_spectralbatch.npz: --norm_D spectralbatch / --norm_E spectralbatch (normalization.py:38-39: BatchNorm2d(affine=True)
    behind the spectral-normed convs): train-mode forward (batch statistics, running buffers updated), the gradient w.r.t. the
    fake image / the style images and every parameter gradient."""
    opt = ref_opt(ndf=8, ngf=8, crop_size=256, norm_D='spectralbatch', norm_E='spectralbatch')
    netD = MultiscaleDiscriminator(opt)
    manD = load_filled(netD)
    label = syn.ellipse_labels(2, 64, 64, seed=71)
    seg = onehot(label)
    fake = torch.from_numpy(syn.smooth_images('d_fake', (2, 1, 64, 64), seed=71)).requires_grad_(True)
    real = torch.from_numpy(syn.smooth_images('d_real', (2, 1, 64, 64), seed=71))
    netD.train()
    pred = netD(torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0))
    loss = sum((p[-1] * torch.from_numpy(syn.hash_uniform('dproj%d' % i, tuple(p[-1].shape), seed=71))).sum() for i, p in enumerate(pred))
    loss = loss + sum(f.abs().mean() for p in pred for f in p[:-1])
    loss.backward()
    rec = {'label': label, 'grad_fake': fake.grad.numpy(), 'loss': np.array([loss.item()])}
    for i in range(2):
        rec['pred_%d' % i] = pred[i][-1].detach().numpy()
        rec['feat_%d_1' % i] = checksum(pred[i][1])
    for k, p in netD.named_parameters():
        rec['gradD_' + k] = checksum(p.grad)
    for k, v in netD.state_dict().items():
        if 'running' in k or k.endswith(('weight_u', 'weight_v')):
            rec['bufD_' + k] = v.detach().numpy().copy()
    netE = ConvEncoder(opt)
    manE = load_filled(netE)
    xs = torch.from_numpy(syn.smooth_images('e_style', (3, 1, 256, 256), seed=72)).requires_grad_(True)
    netE.train()
    mu, logvar, feats = netE(xs)
    (mu.sum() + 0.5 * logvar.sum()).backward()
    rec.update({'e_mu': mu.detach().numpy(), 'e_logvar': logvar.detach().numpy(), 'e_grad_x_ck': checksum(xs.grad)})
    for k, p in netE.named_parameters():
        rec['gradE_' + k] = checksum(p.grad)
    np.savez_compressed(os.path.join(args.out, 'de_spectralbatch.npz'), **rec, **manifest_arrays('D', manD), **manifest_arrays('E', manE))
    print('spectralbatch D/E ok', loss.item())


def options_fixture(args):
    """reference_option_defaults.json: every flag of the reference's TrainOptions / TestOptions parsers
    (options/base_options.py, train_options.py, test_options.py) with its default, type, action and choices."""
    import json
    from options.train_options import TrainOptions
    from options.test_options import TestOptions
    out = {}
    for name, cls in (('train', TrainOptions), ('test', TestOptions)):
        parser = cls().initialize(argparse.ArgumentParser())
        out[name] = {a.dest: {'default': a.default, 'flags': a.option_strings,
                              'type': getattr(a.type, '__name__', None) if a.type else None,
                              'action': type(a).__name__, 'choices': list(a.choices) if a.choices else None}
                     for a in parser._actions if a.dest != 'help'}
    json.dump(out, open(os.path.join(args.out, 'reference_option_defaults.json'), 'w'), indent=0, sort_keys=True, default=str)
    print('options ok', {k: len(v) for k, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=HERE)
    ap.add_argument('--full', action='store_true', help='also the ngf=64 256x256 N=8 pin (slow)')
    ap.add_argument('--only', default='', help="'style': write only trainer_style_ngf8_256.npz (the optional-loss fixture); 'options': only reference_option_defaults.json; 'bn': only g_bn_ngf8_64.npz; 'openeds': only openeds_metric.npz; 'full-train': only trainer_ngf64_256_n8.npz (config 3 as benched, minutes); 'cfg5': only cfg5_ngf64_640x384_n4.npz; 'more': only g_more_ngf8_128.npz; 'cfg5-train': only trainer_ngf64_640x384_n4.npz (one reference G+D step at 640x384, N=4: minutes); 'max': only trainer_max_ngf8_256.npz (--style_aggr_method max)")
    args = ap.parse_args()
    install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    from models.networks.generator import SPADESTYLEGenerator
    from models.networks.discriminator import MultiscaleDiscriminator
    from models.networks.encoder import ConvEncoder
    from models.networks.normalization import SPADE, ApplyStyle, SPADE_STYLE_Block
    from models.networks.architecture import SPADE_STYLE_ResnetBlock
    from models.networks.loss import GANLoss
    from seg2eye_amd import synthetic as syn

    def onehot(label):
        lab = torch.from_numpy(label.astype(np.int64))
        return torch.zeros(lab.shape[0], 4, *lab.shape[2:]).scatter_(1, lab, 1.0)

    if args.only == 'style':
        style_trainer_fixture(args, syn)
        return
    if args.only == 'options':
        options_fixture(args)
        return
    if args.only == 'openeds':
        openeds_fixture(args, syn)
        return
    if args.only == 'bn':
        bn_generator_fixture(args, syn, onehot, SPADESTYLEGenerator)
        return
    if args.only == 'full-train':
        full_train_fixture(args, syn)
        return
    if args.only == 'cfg5-train':
        cfg5_train_fixture(args, syn)
        return
    if args.only == 'max':
        max_aggr_fixture(args, syn)
        return
    if args.only == 'cfg5':
        cfg5_fixture(args, syn, onehot, SPADESTYLEGenerator, MultiscaleDiscriminator)
        return
    if args.only == 'dbatch':
        batchnorm_de_fixture(args, syn, onehot, MultiscaleDiscriminator, ConvEncoder)
        return
    if args.only == 'resize':
        resize_fixture(args, syn)
        return
    if args.only == 'more':
        more_fixture(args, syn, onehot, SPADESTYLEGenerator)
        return

    # ---- G1/G2: generator, ngf=8 (64x64) and ngf=16 (128x128 portrait-ish 128x64) -------
    for tag, ngf, crop, ar, n in (('g_ngf8_64', 8, 64, 1.0, 2), ('g_ngf16_128x64', 16, 64, 0.5, 2)):
        opt = ref_opt(ngf=ngf, crop_size=crop, aspect_ratio=ar)
        netG = SPADESTYLEGenerator(opt)
        man = load_filled(netG)
        H, W = netG.sh * 32, netG.sw * 32
        label = syn.ellipse_labels(n, H, W, seed=7)
        w = syn.hash_normal('latent_w', (n, opt.w_dim), seed=7)
        seg = onehot(label)
        netG.eval()
        with torch.no_grad():
            y_eval = netG(seg, torch.from_numpy(w))
        # gradients of a scalar loss in eval mode (pins backward through every block);
        # taken BEFORE the train-mode forward, which mutates weight_u / weight_v
        netG.zero_grad()
        wt = torch.from_numpy(w).requires_grad_(True)
        yg = netG(seg, wt)
        proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(yg.shape), seed=7))
        (yg * proj).sum().backward()
        grads = {('grad_' + k): checksum(p.grad) for k, p in netG.named_parameters()}
        netG.train()
        with torch.no_grad():
            y_train = netG(seg, torch.from_numpy(w))
        sd_after = netG.state_dict()
        uv = {('uv_' + k): v.numpy().copy() for k, v in sd_after.items() if k.endswith(('weight_u', 'weight_v'))}
        np.savez_compressed(os.path.join(args.out, tag + '.npz'), label=label, w=w,
                            y_eval=y_eval.numpy(), y_train=y_train.numpy(), grad_w=wt.grad.numpy(),
                            hw=np.array([H, W, netG.sh, netG.sw]), **manifest_arrays('G', man), **uv, **grads)
        print(tag, 'y std', float(y_eval.std()), 'train-eval maxdiff', float((y_eval - y_train).abs().max()))

    # ---- G3: per-module ------------------------------------------------------------------
    opt = ref_opt(ngf=8, crop_size=64)
    n, C, h = 2, 16, 16
    label = syn.ellipse_labels(n, 64, 64, seed=11)
    seg = onehot(label)
    x = torch.from_numpy(syn.hash_normal('mod_x', (n, C, h, h), seed=11))
    w = torch.from_numpy(syn.hash_normal('mod_w', (n, 16), seed=11))
    out = {'label': label, 'x': x.numpy(), 'w': w.numpy()}
    for name, ctor in (('spade', lambda: SPADE('spadeinstance3x3', C, 4)),
                       ('adain', lambda: ApplyStyle(16, C, False)),
                       ('ssb', lambda: SPADE_STYLE_Block(C, opt)),
                       ('res_same', lambda: SPADE_STYLE_ResnetBlock(C, C, opt)),
                       ('res_diff', lambda: SPADE_STYLE_ResnetBlock(C, C // 2, opt))):
        m = ctor()
        man = load_filled(m)
        m.eval()
        xi = x.clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        if name == 'spade':
            y = m(xi, seg)
        elif name == 'adain':
            y = m(xi, wi)
        else:
            y = m(xi, seg, wi)
        proj = torch.from_numpy(syn.hash_uniform('proj_' + name, tuple(y.shape), seed=11))
        (y * proj).sum().backward()
        out.update(manifest_arrays(name, man))
        out[name + '_y'] = y.detach().numpy()
        out[name + '_dx'] = xi.grad.numpy()
        if wi.grad is not None:
            out[name + '_dw'] = wi.grad.numpy()
        for k, p in m.named_parameters():
            out['%s_grad_%s' % (name, k)] = checksum(p.grad)
    np.savez_compressed(os.path.join(args.out, 'modules.npz'), **out)
    print('modules ok')

    # ---- D1 + L1: discriminator + losses --------------------------------------------------
    opt = ref_opt(ndf=8, crop_size=32)
    netD = MultiscaleDiscriminator(opt)
    man = load_filled(netD)
    n2, H = 4, 32
    label = syn.ellipse_labels(n2 // 2, H, H, seed=13)
    seg = onehot(label)
    fake = torch.from_numpy(syn.smooth_images('d_fake', (n2 // 2, 1, H, H), seed=13))
    real = torch.from_numpy(syn.smooth_images('d_real', (n2 // 2, 1, H, H), seed=13))
    fake_r = fake.clone().requires_grad_(True)
    xin = torch.cat([torch.cat([seg, fake_r], 1), torch.cat([seg, real], 1)], 0)
    netD.eval()
    pred = netD(xin)
    crit = GANLoss('hinge', tensor=torch.FloatTensor, opt=opt)
    pf = [[t[:t.size(0) // 2] for t in p] for p in pred]
    pr = [[t[t.size(0) // 2:] for t in p] for p in pred]
    l_g = crit(pf, True, for_discriminator=False)
    l_df = crit(pf, False, for_discriminator=True)
    l_dr = crit(pr, True, for_discriminator=True)
    feat = torch.zeros(1)
    for i in range(2):
        for j in range(len(pf[i]) - 1):
            feat = feat + torch.nn.L1Loss()(pf[i][j], pr[i][j].detach()) * 10.0 / 2
    (l_g + feat).sum().backward(retain_graph=True)
    g_fake = fake_r.grad.clone()
    g_params_G = {('gradG_' + k): checksum(p.grad) for k, p in netD.named_parameters()}
    netD.zero_grad()
    (l_df + l_dr).sum().backward()
    g_params_D = {('gradD_' + k): checksum(p.grad) for k, p in netD.named_parameters()}
    netD.train()
    with torch.no_grad():
        pred_t = netD(xin.detach())
    uv = {('uv_' + k): v.numpy() for k, v in netD.state_dict().items() if k.endswith(('weight_u', 'weight_v'))}
    dd = {'label': label, 'fake': fake.numpy(), 'real': real.numpy(),
          'l_g': l_g.detach().numpy(), 'l_df': l_df.detach().numpy(), 'l_dr': l_dr.detach().numpy(),
          'l_feat': feat.detach().numpy(), 'grad_fake': g_fake.numpy()}
    for i in range(2):
        for j in range(5):
            dd['pred_%d_%d' % (i, j)] = pred[i][j].detach().numpy() if j in (0, 4) or i == 1 else checksum(pred[i][j])
            dd['predtrain_%d_%d' % (i, j)] = pred_t[i][j].numpy() if j == 4 else checksum(pred_t[i][j])
    np.savez_compressed(os.path.join(args.out, 'd_ndf8_32.npz'), **dd, **manifest_arrays('D', man),
                        **uv, **g_params_G, **g_params_D)
    print('D ok', [float(x) for x in (l_g, l_df, l_dr, feat)])

    # ---- E1: encoder ----------------------------------------------------------------------
    opt = ref_opt(ngf=8, crop_size=256)
    netE = ConvEncoder(opt)
    man = load_filled(netE)
    xs = syn.smooth_images('e_style', (3, 1, 64, 96), seed=17)       # exercises the bilinear resize
    netE.eval()
    with torch.no_grad():
        mu, logvar, feats = netE(torch.from_numpy(xs))
    np.savez_compressed(os.path.join(args.out, 'e_ngf8.npz'), x=xs, mu=mu.numpy(), logvar=logvar.numpy(),
                        feat_last=feats[-1].numpy(), feat0_ck=checksum(feats[0]), **manifest_arrays('E', man))
    print('E ok')

    # ---- M1 + T1: full model / trainer, crop 256, ngf=ndf=8, N=2 -----------------------------
    class FloatAdam(torch.optim.Adam):           # betas=(0, 0.9) mixes int and float
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatAdam
    from trainers.pix2pix_trainer import Pix2PixTrainer
    opt = ref_opt(ngf=8, ndf=8, crop_size=256, aspect_ratio=1.0, batchSize=2, init_type='normal')
    opt.checkpoints_dir = '/tmp/s2e_golden_ckpt'
    trainer = Pix2PixTrainer(opt)
    model = trainer.pix2pix_model
    mG, mD, mE = load_filled(model.netG), load_filled(model.netD), load_filled(model.netE)
    batch = syn.make_batch(2, 256, 256, seed=21)

    def tdata():
        return {'label': torch.from_numpy(batch['label'].astype(np.int64)),
                'style_image': torch.from_numpy(batch['style_image']),
                'target': torch.from_numpy(batch['target']), 'filename': batch['filename']}
    rec = {}
    for it in range(2):
        trainer.run_generator_one_step(tdata())
        for k, v in trainer.g_losses.items():
            rec['it%d_%s' % (it, k.replace('/', '_'))] = v.detach().numpy()
        if it == 0:
            rec['it0_fake_sub'] = trainer.generated.detach()[:, :, ::8, ::8].numpy()
        trainer.run_discriminator_one_step(tdata())
        for k, v in trainer.d_losses.items():
            rec['it%d_%s' % (it, k.replace('/', '_'))] = v.detach().numpy()
        for tag, net in (('G', model.netG), ('D', model.netD), ('E', model.netE)):
            for k, v in net.state_dict().items():
                rec['it%d_ck_%s.%s' % (it, tag, k)] = checksum(v)
    np.savez_compressed(os.path.join(args.out, 'trainer_ngf8_256.npz'), **rec,
                        **manifest_arrays('G', mG), **manifest_arrays('D', mD), **manifest_arrays('E', mE))
    print('trainer ok', {k: float(v) for k, v in rec.items() if k.startswith('it') and v.size == 1})

    # ---- full-size pin (config 2): ngf=64, 256^2, N=8, eval-mode G --------------------------
    if args.full:
        opt = ref_opt(ngf=64, crop_size=256, aspect_ratio=1.0)
        netG = SPADESTYLEGenerator(opt)
        man = load_filled(netG)
        label = syn.ellipse_labels(8, 256, 256, seed=1234)
        w = syn.hash_normal('latent_w', (8, 16), seed=1234)
        netG.eval()
        with torch.no_grad():
            y = netG(onehot(label), torch.from_numpy(w))
        np.savez_compressed(os.path.join(args.out, 'g_ngf64_256_pin.npz'),
                            y_sub=y[:, :, ::8, ::8].numpy().astype(np.float32),
                            stats=np.array([y.mean().item(), y.std().item(), y.norm().item()]),
                            **manifest_arrays('G', man))
        print('full pin ok: mean/std', y.mean().item(), y.std().item())


if __name__ == '__main__':
    main()
