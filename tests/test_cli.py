"""train.py / test.py surface: option names, types and defaults against the REAL reference's parsers
(tests/golden/reference_option_defaults.json, dumped by make_golden.py's environment), the iteration counter's
arithmetic and resume record, the synthetic data contract; on a GPU: a tiny end-to-end train -> resume -> test run."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

DELIBERATE = {'dataset_mode': 'synthetic'}               # reference: 'openeds' (H5 dataset)


@pytest.mark.parametrize('mode', ['train', 'test'])
def test_cli_flags_match_the_reference(mode):
    from seg2eye_amd.options import build_parser
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_option_defaults.json')))[mode]
    ours = {a.dest: a for a in build_parser(mode == 'train')._actions if a.dest != 'help'}
    missing = sorted(set(ref) - set(ours))
    assert not missing, 'reference flags without a counterpart: %s' % missing
    for name, r in ref.items():
        a = ours[name]
        want = DELIBERATE.get(name, r['default'])
        got = a.default
        if isinstance(want, str) and want in ('inf',):
            want = float('inf')
        assert got == want or (isinstance(want, (int, float)) and float(got) == float(want)), (name, got, want)
        assert (type(a).__name__ == '_StoreTrueAction') == (r['action'] == '_StoreTrueAction'), name
        if r['type']:
            assert a.type.__name__ == r['type'], (name, a.type, r['type'])
        if r['choices']:
            assert list(a.choices) == r['choices'], name


def test_parse_postprocessing():
    from seg2eye_amd.options import parse
    o = parse(['--gpu_ids', '0,1', '--label_nc', '4', '--lambda_l2', '15'])
    assert o.gpu_ids == [0, 1] and o.semantic_nc == 4 and o.isTrain and o.lambda_l2 == 15.0 and o.aspect_ratio == 0.8
    t = parse(['--produce_npy'], is_train=False)
    assert not t.isTrain and t.produce_npy and t.results_dir == 'results/' and t.lambda_feat == 10.0


def test_iteration_counter_matches_reference_arithmetic(tmp_path):
    from seg2eye_amd.iter_counter import IterationCounter
    from seg2eye_amd.options import parse
    o = parse(['--name', 'r', '--checkpoints_dir', str(tmp_path), '--batchSize', '4', '--print_freq', '8',
               '--save_latest_freq', '16', '--niter', '2', '--niter_decay', '1'])
    c = IterationCounter(o, 32)
    assert list(c.training_epochs()) == [1, 2, 3]
    hits = []
    for i in range(8):
        c.record_one_iteration()
        hits.append((c.total_steps_so_far, c.needs_printing(), c.needs_saving()))
    assert [h[0] for h in hits] == [4, 8, 12, 16, 20, 24, 28, 32]
    assert [h[1] for h in hits] == [False, True, False, True, False, True, False, True]      # steps counted in samples
    assert [h[2] for h in hits] == [False, False, False, True, False, False, False, True]
    c.record_current_iter()
    assert np.loadtxt(c.iter_record_path, delimiter=',', dtype=int).tolist() == [1, 32]
    o2 = parse(['--name', 'r', '--checkpoints_dir', str(tmp_path), '--batchSize', '4', '--continue_train'])
    c2 = IterationCounter(o2, 32)
    assert (c2.first_epoch, c2.epoch_iter, c2.total_steps_so_far) == (1, 32, 32)
    c.record_epoch_end()                                                     # save_epoch_freq 1: next epoch, iter 0
    assert np.loadtxt(c.iter_record_path, delimiter=',', dtype=int).tolist() == [2, 0]


def test_synthetic_data_contract():
    from seg2eye_amd.data import create_dataloader
    from seg2eye_amd.options import parse
    o = parse(['--batchSize', '2', '--crop_size', '64', '--aspect_ratio', '0.5', '--synthetic_size', '8'])
    dl = create_dataloader(o, rank=1, world=2)
    assert len(dl) == 2
    b = next(iter(dl))
    assert b['label'].shape == (2, 1, 128, 64) and b['label'].dtype == torch.uint8 and int(b['label'].max()) <= 3
    assert b['style_image'].shape == (2, 4, 1, 128, 64) and b['target'].shape == (2, 1, 128, 64)
    assert float(b['target'].abs().max()) <= 1.0 and len(b['filename']) == 2
    other = next(iter(create_dataloader(o, rank=0, world=2)))
    assert not torch.equal(other['label'], b['label'])                      # ranks see different shards
    with pytest.raises(NotImplementedError):
        create_dataloader(parse(['--dataset_mode', 'openeds']))


@pytest.mark.gpu
def test_train_resume_test_end_to_end(tmp_path):
    import train as train_mod
    import test as test_mod
    common = ['--name', 'e2e', '--checkpoints_dir', str(tmp_path / 'ck'), '--ngf', '8', '--ndf', '8', '--batchSize', '2',
              '--aspect_ratio', '1.0', '--synthetic_size', '4', '--compute_dtype', 'fp32']
    tr = train_mod.main(common + ['--niter', '1', '--niter_decay', '1', '--print_freq', '2', '--lambda_l2', '15'])
    ck = tmp_path / 'ck' / 'e2e'
    for f in ('latest_net_G.pth', 'latest_net_D.pth', 'latest_net_E.pth', '1_net_G.pth', '2_net_G.pth', 'iter.txt'):
        assert (ck / f).exists(), f
    sd = torch.load(ck / 'latest_net_G.pth')
    assert 'head_0.conv_0.weight_orig' in sd and 'fc.weight' in sd       # reference key names
    w_end = tr.pix2pix_model.netG.fc.weight.detach().cpu().clone()
    assert torch.equal(sd['fc.weight'], w_end)
    tr2 = train_mod.main(common + ['--niter', '2', '--niter_decay', '1', '--continue_train'])    # resumes at epoch 3
    assert not torch.equal(tr2.pix2pix_model.netG.fc.weight.detach().cpu(), w_end)
    targs = ['--name', 'e2e', '--checkpoints_dir', str(tmp_path / 'ck'), '--results_dir', 'res', '--ngf', '8',
             '--batchSize', '2', '--aspect_ratio', '1.0', '--synthetic_size', '4', '--compute_dtype', 'fp32']
    # inference: one (1, 640, 400) uint8 .npy per sample + the list file (util/tester.py:193-219)
    paths = test_mod.main(targs + ['--produce_npy'])
    assert len(paths) == 4
    rdir = tmp_path / 'ck' / 'e2e' / 'res' / 'train'
    assert sorted(os.listdir(rdir)) == sorted([os.path.basename(p) for p in paths] + ['pred_npy_list.txt'])
    img = np.load(paths[0])
    assert img.shape == (1, 640, 400) and img.dtype == np.uint8
    # validation: per-image OpenEDS errors + the x1471 statistic (util/tester.py:99-121,165-176)
    errs, stats = test_mod.main(targs + ['--dataset_key', 'validation'])
    assert len(errs) == 4 and all(0.0 < e < 1.0 for e in errs)
    assert abs(stats['mse/validation/full/relative'] - float(np.mean(errs)) * 1471) < 1e-3
