"""The CPU oracle against the golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, filled_state, assert_checksum_close
from oracle import seg2eye_oracle as O
from seg2eye_amd import synthetic as syn
from seg2eye_amd.options import default_opt

TOL = 2e-5      # fp32 CPU vs fp32 CPU, different op grouping only


def _onehot(label):
    return O.one_hot_labels(torch.from_numpy(label.astype(np.int64)), 4)


@pytest.mark.parametrize('tag', ['g_ngf8_64', 'g_ngf16_128x64'])
def test_generator_eval_train_and_grads(tag):
    z = load_golden(tag)
    sd = filled_state(z, 'G')
    H, W, sh, sw = [int(v) for v in z['hw']]
    seg, w = _onehot(z['label']), torch.from_numpy(z['w'])
    with torch.no_grad():
        y = O.generator_forward(sd, seg, w, sh, sw, training=False)
    np.testing.assert_allclose(y.numpy(), z['y_eval'], atol=TOL, rtol=0)
    upd = {}
    with torch.no_grad():
        yt = O.generator_forward(sd, seg, w, sh, sw, training=True, updates=upd)
    np.testing.assert_allclose(yt.numpy(), z['y_train'], atol=TOL, rtol=0)
    uv_keys = [k[3:] for k in z.files if k.startswith('uv_')]
    assert sorted(uv_keys) == sorted(upd)
    for k in uv_keys:
        np.testing.assert_allclose(upd[k].numpy(), z['uv_' + k], atol=1e-6, rtol=0)
    # backward, eval mode
    leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
    wt = w.clone().requires_grad_(True)
    yg = O.generator_forward(leaf, seg, wt, sh, sw, training=False)
    proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(yg.shape), seed=7))
    (yg * proj).sum().backward()
    np.testing.assert_allclose(wt.grad.numpy(), z['grad_w'], atol=2e-4 * np.abs(z['grad_w']).max(), rtol=0)
    for k, p in leaf.items():
        if O.OracleModel.is_param(k):
            assert_checksum_close(p.grad, z['grad_' + k], 2e-4, k)


def test_modules():
    z = load_golden('modules')
    seg = _onehot(z['label'])
    x0, w0 = torch.from_numpy(z['x']), torch.from_numpy(z['w'])
    fns = {
        'spade': lambda sd, x, w: O.spade(sd, '', x, seg) if False else O.spade({('.' + k): v for k, v in sd.items()}, '', x, seg),
        'adain': lambda sd, x, w: O.apply_style({('p.' + k): v for k, v in sd.items()}, 'p', x, w),
        'ssb': lambda sd, x, w: O.spade_style_block({('p.' + k): v for k, v in sd.items()}, 'p', x, seg, w),
        'res_same': lambda sd, x, w: O.spade_style_resblk({('p.' + k): v for k, v in sd.items()}, 'p', x, seg, w, False, None),
        'res_diff': lambda sd, x, w: O.spade_style_resblk({('p.' + k): v for k, v in sd.items()}, 'p', x, seg, w, False, None),
    }
    for name, fn in fns.items():
        sd = filled_state(z, name)
        leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        y = fn(leaf, x, w)
        np.testing.assert_allclose(y.detach().numpy(), z[name + '_y'], atol=TOL, rtol=0, err_msg=name)
        proj = torch.from_numpy(syn.hash_uniform('proj_' + name, tuple(y.shape), seed=11))
        (y * proj).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), z[name + '_dx'], atol=5e-5, rtol=0, err_msg=name)
        if name + '_dw' in z.files:
            np.testing.assert_allclose(w.grad.numpy(), z[name + '_dw'], atol=2e-4 * np.abs(z[name + '_dw']).max(), rtol=0)
        for k, p in leaf.items():
            if O.OracleModel.is_param(k):
                assert_checksum_close(p.grad, z['%s_grad_%s' % (name, k)], 2e-4, name + ':' + k)


def test_discriminator_and_losses():
    z = load_golden('d_ndf8_32')
    sd = filled_state(z, 'D')
    leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
    seg = _onehot(z['label'])
    fake = torch.from_numpy(z['fake']).requires_grad_(True)
    real = torch.from_numpy(z['real'])
    xin = torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0)
    pred = O.discriminator_forward(leaf, xin, training=False)
    for i in range(2):
        for j in range(5):
            ref = z['pred_%d_%d' % (i, j)]
            if ref.ndim == 1 and ref.shape[0] == 18:
                assert_checksum_close(pred[i][j], ref, 1e-5, 'pred%d%d' % (i, j))
            else:
                np.testing.assert_allclose(pred[i][j].detach().numpy(), ref, atol=TOL, rtol=0)
    pf, pr = O.divide_pred(pred)
    l_g = O.gan_loss(pf, True, for_discriminator=False)
    l_df = O.gan_loss(pf, False, True)
    l_dr = O.gan_loss(pr, True, True)
    feat = O.feature_matching_loss(pf, pr, 10.0)
    for got, key in ((l_g, 'l_g'), (l_df, 'l_df'), (l_dr, 'l_dr'), (feat, 'l_feat')):
        assert got.shape == (1,)
        np.testing.assert_allclose(got.detach().numpy(), z[key], atol=1e-5, rtol=1e-5)
    params = [(k, p) for k, p in leaf.items() if O.OracleModel.is_param(k)]
    grads = torch.autograd.grad((l_g + feat).sum(), [fake] + [p for _, p in params], retain_graph=True)
    np.testing.assert_allclose(grads[0].numpy(), z['grad_fake'], atol=2e-4 * np.abs(z['grad_fake']).max(), rtol=0)
    for (k, _), g in zip(params, grads[1:]):
        assert_checksum_close(g, z['gradG_' + k], 2e-4, k)
    grads = torch.autograd.grad((l_df + l_dr).sum(), [p for _, p in params])
    for (k, _), g in zip(params, grads):
        assert_checksum_close(g, z['gradD_' + k], 2e-4, k)
    # train mode: one power iteration, post-forward u, v
    upd = {}
    with torch.no_grad():
        pt = O.discriminator_forward(sd, xin.detach(), training=True, updates=upd)
    for i in range(2):
        np.testing.assert_allclose(pt[i][4].numpy(), z['predtrain_%d_4' % i], atol=TOL, rtol=0)
    for k in upd:
        np.testing.assert_allclose(upd[k].numpy(), z['uv_' + k], atol=1e-6, rtol=0)


def test_encoder():
    z = load_golden('e_ngf8')
    sd = filled_state(z, 'E')
    with torch.no_grad():
        mu, logvar, feats = O.encoder_forward(sd, torch.from_numpy(z['x']), training=False)
    np.testing.assert_allclose(mu.numpy(), z['mu'], atol=5e-5, rtol=0)
    np.testing.assert_allclose(logvar.numpy(), z['logvar'], atol=5e-5, rtol=0)
    np.testing.assert_allclose(feats[-1].numpy(), z['feat_last'], atol=5e-5, rtol=0)
    assert_checksum_close(feats[0], z['feat0_ck'], 1e-5)


def test_trainer_two_iterations():
    """T1: Pix2PixTrainer G-step, D-step, G-step, D-step (TTUR Adam, double
    power iteration, G regeneration in the D step)."""
    z = load_golden('trainer_ngf8_256')
    opt = default_opt(ngf=8, ndf=8, crop_size=256, aspect_ratio=1.0, batchSize=2)
    m = O.OracleModel(filled_state(z, 'G'), filled_state(z, 'D'), filled_state(z, 'E'), opt, 8, 8)
    b = syn.make_batch(2, 256, 256, seed=21)
    data = {'label': torch.from_numpy(b['label'].astype(np.int64)),
            'style_image': torch.from_numpy(b['style_image']), 'target': torch.from_numpy(b['target'])}
    for it in range(2):
        gl, fake = m.run_generator_one_step(data)
        dl = m.run_discriminator_one_step(data)
        for k, v in list(gl.items()) + list(dl.items()):
            ref = z['it%d_%s' % (it, k.replace('/', '_'))]
            np.testing.assert_allclose(v.numpy(), ref, rtol=2e-4, atol=2e-5, err_msg='it%d %s' % (it, k))
        if it == 0:
            np.testing.assert_allclose(fake[:, :, ::8, ::8].numpy(), z['it0_fake_sub'], atol=1e-4, rtol=0)
        for tag, sd in (('G', m.G), ('D', m.D), ('E', m.E)):
            for k, v in sd.items():
                assert_checksum_close(v, z['it%d_ck_%s.%s' % (it, tag, k)], 1e-4 if it == 0 else 5e-4,
                                      'it%d %s.%s' % (it, tag, k))


STYLE_LAMBDAS = dict(lambda_l2=15.0, lambda_l1=2.0, lambda_style_w=0.5, lambda_style_feat=0.001, lambda_gram=10000.0)


def test_trainer_optional_losses():
    """T2: the reference's own training recipe (L2/L1 + the style-consistency terms that re-encode the generated
    image, pix2pix_model.py:196-229): one G step + one D step against the real reference."""
    z = load_golden('trainer_style_ngf8_256')
    opt = default_opt(ngf=8, ndf=8, crop_size=256, aspect_ratio=1.0, batchSize=2, **STYLE_LAMBDAS)
    m = O.OracleModel(filled_state(z, 'G'), filled_state(z, 'D'), filled_state(z, 'E'), opt, 8, 8)
    b = syn.make_batch(2, 256, 256, seed=23)
    data = {'label': torch.from_numpy(b['label'].astype(np.int64)),
            'style_image': torch.from_numpy(b['style_image']), 'target': torch.from_numpy(b['target'])}
    gl, fake = m.run_generator_one_step(data)
    dl = m.run_discriminator_one_step(data)
    assert set(gl) == {'GAN', 'L2/weighted', 'L1/weighted', 'style_w/weighted', 'style_feat/weighted', 'gram/weighted', 'GAN_Feat'}
    for k, v in list(gl.items()) + list(dl.items()):
        ref = z['it0_%s' % k.replace('/', '_')]
        np.testing.assert_allclose(v.numpy().reshape(-1), ref, rtol=2e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(fake[:, :, ::8, ::8].numpy(), z['it0_fake_sub'], atol=1e-4, rtol=0)
    for tag, sd in (('G', m.G), ('D', m.D), ('E', m.E)):
        for k, v in sd.items():
            assert_checksum_close(v, z['it0_ck_%s.%s' % (tag, k)], 1e-4, '%s.%s' % (tag, k))


def test_trainer_max_aggregation():
    """T4: --style_aggr_method max (pix2pix_model.py:271-278): the style codes, one G step + one D step, and netE's parameter
    gradients (which reach only the arg-max style image per component) against the real reference."""
    z = load_golden('trainer_max_ngf8_256')
    opt = default_opt(ngf=8, ndf=8, crop_size=256, aspect_ratio=1.0, batchSize=2, style_aggr_method='max')
    m = O.OracleModel(filled_state(z, 'G'), filled_state(z, 'D'), filled_state(z, 'E'), opt, 8, 8)
    b = syn.make_batch(2, 256, 256, seed=29)
    data = {'label': torch.from_numpy(b['label'].astype(np.int64)),
            'style_image': torch.from_numpy(b['style_image']), 'target': torch.from_numpy(b['target'])}
    with torch.no_grad():
        w = O.encode_w(dict(m.E), data['style_image'], 'max', training=True)
    np.testing.assert_allclose(w.numpy(), z['w'], atol=2e-5, rtol=1e-4)
    w_mean = O.encode_w(dict(m.E), data['style_image'], 'mean', training=True)
    assert float((w - w_mean).abs().max()) > 1e-3                       # (the fixture does tell the two aggregations apart)
    # netE's gradients of the G step (what run_generator_one_step differentiates, before its Adam update)
    G, D, E = m._leaf(m.G), m._leaf(m.D), m._leaf(m.E)
    losses, _ = m.generator_losses(G, D, E, data, True, ({}, {}, {}))
    names = [k for k in E if m.is_param(k)]
    grads = torch.autograd.grad(sum(losses.values()).mean(), [E[k] for k in names], allow_unused=True)
    seen = 0
    for k, g in zip(names, grads):
        key = 'it0_grad_E.' + k
        assert (g is not None) == (key in z.files), k
        if g is not None:
            assert_checksum_close(g, z[key], 5e-4, 'grad E.' + k)
            seen += 1
    assert seen >= 8
    gl, fake = m.run_generator_one_step(data)
    dl = m.run_discriminator_one_step(data)
    for k, v in list(gl.items()) + list(dl.items()):
        np.testing.assert_allclose(v.numpy().reshape(-1), z['it0_%s' % k.replace('/', '_')], rtol=2e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(fake[:, :, ::8, ::8].numpy(), z['it0_fake_sub'], atol=1e-4, rtol=0)
    for tag, sd in (('G', m.G), ('D', m.D), ('E', m.E)):
        for k, v in sd.items():
            assert_checksum_close(v, z['it0_ck_%s.%s' % (tag, k)], 1e-4, '%s.%s' % (tag, k))


def test_generator_batchnorm_spade():
    """G3: the reference's default --norm_G spectralspadebatch3x3: train-mode forward (batch statistics), every parameter
    gradient, the updated BatchNorm running buffers and u/v, then eval mode on the updated buffers."""
    z = load_golden('g_bn_ngf8_64')
    sd = filled_state(z, 'G')
    seg, w = _onehot(z['label']), torch.from_numpy(z['w'])
    leaf = {k: (v.clone().requires_grad_(True) if O.OracleModel.is_param(k) else v) for k, v in sd.items()}
    wt = w.clone().requires_grad_(True)
    upd = {}
    y = O.generator_forward(leaf, seg, wt, 2, 2, training=True, updates=upd)
    np.testing.assert_allclose(y.detach().numpy(), z['y_train'], atol=TOL, rtol=0)
    proj = torch.from_numpy(syn.hash_uniform('g_proj', tuple(y.shape), seed=9))
    (y * proj).sum().backward()
    np.testing.assert_allclose(wt.grad.numpy(), z['grad_w'], atol=2e-4 * np.abs(z['grad_w']).max(), rtol=0)
    for k, p in leaf.items():
        if O.OracleModel.is_param(k):
            assert_checksum_close(p.grad, z['grad_' + k], 2e-4, k)
    bufs = [k[4:] for k in z.files if k.startswith('buf_')]
    assert sorted(bufs) == sorted(upd)
    for k in bufs:
        np.testing.assert_allclose(upd[k].detach().numpy(), z['buf_' + k], atol=1e-6, rtol=0, err_msg=k)
    with torch.no_grad():
        ye = O.generator_forward({**sd, **{k: v.detach() for k, v in upd.items()}}, seg, w, 2, 2, training=False)
    np.testing.assert_allclose(ye.numpy(), z['y_eval_after'], atol=TOL, rtol=0)


def test_openeds_metric():
    """the oracle's restatement of the OpenEDS error (loss.py:102-171, postprocessor.py:58-97) against the
    REAL reference's outputs on synthetic images regenerated from the fixture's seeds (tests/golden/openeds_metric.npz)."""
    z = load_golden('openeds_metric')
    sa, sb, sia, sib = (int(x) for x in z['seeds'])
    a = torch.from_numpy(syn.make_batch(3, 96, 80, seed=sa)['target'])
    b = torch.from_numpy(syn.make_batch(3, 96, 80, seed=sb)['target'])
    assert np.array_equal(O.to_255(a).numpy().astype(np.uint8), z['to255_a'])                  # integer work: bit-exact
    np.testing.assert_allclose(O.mse_for_tensors(a, b).numpy(), z['mse_tensors'], rtol=1e-6)
    ia = O.to_255(torch.from_numpy(syn.make_batch(2, 640, 400, seed=sia)['target']))
    ib = O.to_255(torch.from_numpy(syn.make_batch(2, 640, 400, seed=sib)['target']))
    assert [int(ia.long().sum()), int((ia.long() * ia.long()).sum())] == z['ia_checksum'].tolist()
    np.testing.assert_allclose(O.mse_for_images(ia, ib).numpy(), z['mse_images'], rtol=1e-6)
    np.testing.assert_allclose(float(O.openeds_accuracy(ia[0], ib[0])), float(z['acc_single']), rtol=1e-6)
    st = O.error_statistics(z['stat_errors'], 'full', 'validation')
    assert list(st.keys()) == [str(z['stat_key'])]
    np.testing.assert_allclose(list(st.values())[0], float(z['stat_value']), rtol=1e-12)
    # the resize restatement (cv2 absent: parity-unpinned, see the oracle): shape, range and exactness on constants
    r = O.to_255_resized(a)
    assert r.shape == (3, 1, 640, 400) and int(r.min()) >= 0 and int(r.max()) <= 255
    assert int(O.to_255_resized(torch.full((1, 1, 8, 8), 0.5)).unique().item()) == int((0.5 + 1) * 255 / 2)



def test_resize_rule_matches_fixture():
    """f3: the oracle's explicit float64 cv2.INTER_LINEAR restatement + unnormalize against the committed vectors (generated by
    `make_golden.py --only resize`, which also ran the REFERENCE's own `ImageProcessor.unnormalize` on the resized float64 image:
    the truncation step is the reference's code, the resize rule is OpenCV's published algorithm -- cv2 is not installable here).
    Also the rule's defining properties: identity at equal size, edge clamp, exact 2x upsampling taps."""
    from seg2eye_amd import synthetic as syn
    z = load_golden('resize_cv2_rule')
    for i, seed in enumerate(z['seeds']):
        H, W = (int(v) for v in z['hw_%d' % i])
        img = torch.from_numpy(syn.make_batch(1, H, W, seed=int(seed))['target'])
        pre = O.to_255_pre_truncation(img)
        assert pre.dtype == torch.float64 and tuple(pre.shape) == (1, 1, 640, 400)
        np.testing.assert_array_equal(pre[0, 0, ::7, ::5].numpy(), z['pre_sub_%d' % i])             # bit-exact float64
        q = O.to_255_resized(img)
        np.testing.assert_array_equal(q[0, 0, ::3, ::3].numpy().astype(np.uint8), z['q_sub_%d' % i])
        assert [int(q.long().sum()), int((q.long() ** 2).sum())] == [int(v) for v in z['q_sums_%d' % i]]
    x = torch.from_numpy(syn.make_batch(1, 12, 10, seed=3)['target'])
    assert torch.equal(O.resize_bilinear(x, w=10, h=12), x.double())                                 # same size: identity
    up = O.resize_bilinear(x, w=20, h=12)[0, 0]                                                      # 2x along x: taps (0.75, 0.25)
    xd = x.double()[0, 0]
    assert torch.equal(up[:, 0], xd[:, 0]) and torch.equal(up[:, -1], xd[:, -1])                     # edge clamp
    np.testing.assert_allclose(up[:, 1].numpy(), (0.75 * xd[:, 0] + 0.25 * xd[:, 1]).numpy(), rtol=0, atol=1e-15)
    np.testing.assert_allclose(up[:, 2].numpy(), (0.25 * xd[:, 0] + 0.75 * xd[:, 1]).numpy(), rtol=0, atol=1e-15)
