"""Option namespace for the Seg2Eye hot path.

Mirrors the flag names and defaults the reference parses in
``options/base_options.py:21-64``, ``options/train_options.py:13-51``,
``models/networks/generator.py:15-20`` and
``models/networks/discriminator.py:16-28,69-72`` -- only the flags that shape
the G+D train step.  One deliberate difference: ``norm_G`` defaults to the
InstanceNorm variant (``spectralspadeinstance3x3``) because that is what the
hot path (BASELINE.json north_star) is defined on; the BatchNorm
SPADE variant is not built and asking for it raises.
"""
import argparse

_DEFAULTS = dict(
    name='seg2eye_amd', gpu_ids=[0], checkpoints_dir='./checkpoints', model='pix2pix',
    norm_G='spectralspadeinstance3x3', norm_D='spectralinstance', norm_E='spectralinstance',
    netG='spadestyle', netD='multiscale', netE='conv',
    batchSize=1, preprocess_mode='fixed', load_size=256, crop_size=256, aspect_ratio=1.0,
    label_nc=4, input_nc=1, output_nc=1, input_ns=4, semantic_nc=4,
    style_aggr_method='mean', style_sample_method='random',
    ngf=64, ndf=64, nef=16, w_dim=16, init_type='xavier', init_variance=0.02,
    isTrain=True, continue_train=False, which_epoch='latest',
    niter=14, niter_decay=7, optimizer='adam', beta1=0.5, beta2=0.999, lr=2e-4,
    D_steps_per_G=1, weight_decay=0.0,
    lambda_feat=10.0, lambda_vgg=10.0, lambda_l2=0.0, lambda_l1=0.0, lambda_openeds=0.0,
    lambda_kld=0.05, lambda_style_w=0.0, lambda_style_feat=0.0, lambda_gram=0.0,
    no_ganFeat_loss=False, no_vgg_loss=True, gan_mode='hinge', no_TTUR=False,
    num_upsampling_layers='normal', netD_subarch='n_layer', num_D=2, n_layers_D=4,
    # build-only knobs (no reference counterpart)
    compute_dtype='bf16',      # 'bf16' | 'fp32': storage + MFMA input type of the HIP path
    hip_graphs=False,          # capture zero_grad+forward+backward of each step into a hipGraph (fixed shapes)
)


def default_opt(**overrides):
    """Return an ``argparse.Namespace`` with the reference's field list
     and the given overrides."""
    d = dict(_DEFAULTS)
    unknown = set(overrides) - set(d)
    if unknown:
        raise KeyError('unknown option(s): %s' % sorted(unknown))
    d.update(overrides)
    d['semantic_nc'] = d['label_nc']          # options/base_options.py:150
    if isinstance(d['gpu_ids'], str):         # options/base_options.py:153-158
        d['gpu_ids'] = [int(s) for s in d['gpu_ids'].split(',') if int(s) >= 0]
    return argparse.Namespace(**d)


def latent_size(opt):
    """(sw, sh) of the generator's starting feature map,
    models/networks/generator.py:52-67 (python ``round`` = banker's)."""
    if opt.num_upsampling_layers == 'normal':
        n_up = 5
    elif opt.num_upsampling_layers == 'more':
        n_up = 6
    else:
        # 'most' is broken in the reference (generator.py:44-46)
        raise ValueError('opt.num_upsampling_layers [%s] not supported' % opt.num_upsampling_layers)
    sw = opt.crop_size // (2 ** n_up)
    sh = round(sw / opt.aspect_ratio)
    return sw, sh


def image_hw(opt):
    """Legal (H, W) of labels/images for this opt."""
    sw, sh = latent_size(opt)
    f = 32 if opt.num_upsampling_layers == 'normal' else 64
    return sh * f, sw * f
