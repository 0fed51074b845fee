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
        create_dataloader(parse(['--dataset_mode', 'cityscapes']))


@pytest.mark.gpu
def test_train_resume_test_end_to_end(tmp_path):
    import train as train_mod
    import test as test_mod
    common = ['--name', 'e2e', '--checkpoints_dir', str(tmp_path / 'ck'), '--ngf', '8', '--ndf', '8', '--batchSize', '2',
              '--aspect_ratio', '1.0', '--synthetic_size', '4', '--compute_dtype', 'fp32']
    tr = train_mod.main(common + ['--niter', '1', '--niter_decay', '1', '--print_freq', '2', '--lambda_l2', '15', '--lambda_openeds', '1',
                                  '--display_freq', '4', '--full_val_freq', '4', '--validation_limit', '2'])   # with the periodic validation passes
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



@pytest.mark.gpu
def test_two_rank_train_with_batchnorm_spade_and_rank0_validation(tmp_path):
    """with the CLI's default --norm_G spectralspadebatch3x3 the SPADE layers exchange their batch statistics
    between replicas in every training forward / backward.  Rank 0's periodic validation passes run ALONE (train mode, as in
    the reference) while the other ranks wait in the buffer broadcast that follows: inside them no collective may be issued
    (distributed.solo), or it would pair up with that broadcast.  Two ranks sharing this GPU over gloo run train.py with
    validation due every iteration; the run must finish (a mismatched collective hangs or corrupts) and leave checkpoints."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, S2E_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'train.py'), '--name', 'dp', '--checkpoints_dir', str(tmp_path / 'ck'),
           '--ngf', '8', '--ndf', '8', '--batchSize', '2', '--aspect_ratio', '1.0', '--synthetic_size', '8', '--compute_dtype', 'fp32',
           '--norm_G', 'spectralspadebatch3x3', '--niter', '1', '--niter_decay', '0', '--print_freq', '2', '--display_freq', '2',
           '--full_val_freq', '4', '--validation_limit', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert 'Training was successfully finished.' in out.stdout
    assert (tmp_path / 'ck' / 'dp' / 'latest_net_G.pth').exists()


def _fake_openeds_store(seed=0, users=('U001', 'U002', 'U003'), n_ss=(3, 2, 4), n_gen=5):
    """An in-memory stand-in for the OpenEDS H5 file (data/prepare_openeds.py:77-138): /<key>/<user>/{images_ss, labels_ss,
    images_gen, images_seq, labels_gen} uint8 (n, 640, 400) + *_filenames (S13)."""
    rng = np.random.RandomState(seed)
    store = {}
    for key in ('train', 'validation', 'test'):
        store[key] = {}
        for u, n in zip(users, n_ss):
            g = {'images_ss': rng.randint(0, 256, (n, 640, 400)).astype(np.uint8),
                 'labels_ss': rng.randint(0, 4, (n, 640, 400)).astype(np.uint8),
                 'images_gen': rng.randint(0, 256, (n_gen, 640, 400)).astype(np.uint8),
                 'images_seq': rng.randint(0, 256, (2, 640, 400)).astype(np.uint8),
                 'labels_gen': rng.randint(0, 4, (n, 640, 400)).astype(np.uint8),
                 'images_ss_filenames': np.array([('%s.%03d_ss' % (u, i)).encode() for i in range(n)], dtype='S13'),
                 'labels_gen_filenames': np.array([('%s_%03d_gen' % (u, i)).encode() for i in range(n)], dtype='S13')}
            store[key][u] = g
    return store


def test_openeds_dataset_semantics():
    """the reference's dataset behaviour (data/openeds_dataset.py:40-209, base_dataset.py:51-146) on an
    in-memory H5-like store: flat index -> (user, index), key sets of train/validation vs test, 'fixed' preprocessing
    (nearest for masks, PIL bicubic for images, [-1, 1]), flip augmentation, style sampling modes, the batch contract."""
    from seg2eye_amd.data import create_dataloader
    from seg2eye_amd.openeds_dataset import OpenEDSDataset, resize_nearest
    from seg2eye_amd.options import parse
    from PIL import Image
    store = _fake_openeds_store()
    o = parse(['--dataset_mode', 'openeds', '--dataset_key', 'validation', '--crop_size', '64', '--aspect_ratio', '0.8',
               '--style_sample_method', 'first', '--no_flip', '--batchSize', '2', '--serial_batches'])
    ds = OpenEDSDataset(o, store=store)
    assert len(ds) == 9 and ds.N_start == [0, 3, 5]
    assert ds._get_tuple_identifier_from_index(4) == ('U002', 1) and ds._get_tuple_identifier_from_index(8) == ('U003', 3)
    assert sorted(ds.get_validation_indices()) == [0, 2, 3, 4, 5, 8]                      # first and last sample of every person
    it = ds[4]
    assert it['user'] == 'U002' and it['filename'] == 'U002001_ss'                        # the dot is removed
    assert it['label'].shape == (80, 64) and it['label'].dtype == torch.uint8              # h = round(64 / 0.8)
    assert torch.equal(it['label'], torch.from_numpy(resize_nearest(store['validation']['U002']['labels_ss'][1], 64, 80)))
    assert it['style_image'].shape == (4, 1, 80, 64) and it['target'].shape == (1, 80, 64)
    assert float(it['target'].min()) >= -1 and float(it['target'].max()) <= 1
    ref = np.asarray(Image.fromarray(store['validation']['U002']['images_ss'][1], mode='L').resize((64, 80), Image.BICUBIC), dtype=np.float32)
    assert torch.allclose(it['target'][0], torch.from_numpy(ref / 255.0 - 0.5) / 0.5, atol=1e-6)
    assert it['target_original'].shape == (1, 640, 400) and it['target_original'].dtype == torch.int32
    first = np.asarray(Image.fromarray(store['validation']['U002']['images_gen'][0], mode='L').resize((64, 80), Image.BICUBIC), dtype=np.float32)
    assert torch.allclose(it['style_image'][0, 0], torch.from_numpy(first / 255.0 - 0.5) / 0.5, atol=1e-6)     # 'first': images_gen[0..3]
    # test split: other keys, no target
    ot = parse(['--dataset_mode', 'openeds', '--dataset_key', 'test', '--crop_size', '64', '--aspect_ratio', '0.8', '--no_flip'], is_train=False)
    t = OpenEDSDataset(ot, store=store)[0]
    assert 'target' not in t and t['filename'] == 'U001_000_gen' and t['style_image'].shape == (4, 1, 80, 64)
    # flip augmentation flips mask, images and the ORIGINAL consistently (train mode, flip drawn True)
    class _Always:                       # rng stub: random() -> 0.9 (> 0.5: flip), choice -> first n
        def random(self): return 0.9
        def choice(self, seq, n): return list(seq)[:n]
    otr = parse(['--dataset_mode', 'openeds', '--dataset_key', 'train', '--crop_size', '64', '--aspect_ratio', '0.8', '--style_sample_method', 'random'])
    f = OpenEDSDataset(otr, store=store, rng=_Always())[0]
    g = OpenEDSDataset(parse(['--dataset_mode', 'openeds', '--dataset_key', 'train', '--crop_size', '64', '--aspect_ratio', '0.8',
                              '--style_sample_method', 'random', '--no_flip']), store=store, rng=_Always())[0]
    assert torch.equal(f['label'], g['label'].flip(-1)) and torch.allclose(f['target'], g['target'].flip(-1))
    assert torch.equal(f['target_original'], g['target_original'].flip(-1)) and torch.allclose(f['style_image'], g['style_image'].flip(-1))
    # ref sampling: a similarity ranking per (user, filename), optionally pointing into the sequence images
    refs = {'train': {'U001': {'U001000_ss': {'index': np.array([4, 6, 0, 2, 1]), 'subset': np.array([b'g', b's', b'g', b'g', b'g'])}}}}
    orf = parse(['--dataset_mode', 'openeds', '--dataset_key', 'train', '--crop_size', '64', '--aspect_ratio', '0.8',
                 '--style_sample_method', 'ref_first', '--no_flip'])
    r = OpenEDSDataset(orf, store=store, style_refs=refs)
    st, idx, sub = r.get_style_images('U001', 4, lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()), 'U001000_ss')
    assert idx == [4, 1, 0, 2] and [bytes(x) for x in sub] == [b'g', b's', b'g', b'g']      # 6 - n_images(5) = 1 into images_seq
    assert torch.equal(st[1], torch.from_numpy(store['train']['U001']['images_seq'][1]))
    # the DataLoader contract
    dl = create_dataloader(o, store=store)
    b = next(iter(dl))
    assert dl.N == 9 and b['label'].shape == (2, 80, 64) and b['style_image'].shape == (2, 4, 1, 80, 64)
    assert b['target_original'].shape == (2, 1, 640, 400) and len(b['filename']) == 2


@pytest.mark.gpu
def test_tester_on_openeds_store(tmp_path):
    """SURVEY 8 f3 + f4 together: the Tester walks an OpenEDS-layout store (in memory), generates with the HIP generator,
    resizes / truncates / scores on the device; 'fix' mode uses the dataset's per-person index list, 'full' all samples;
    errors agree with the CPU oracle applied to the same generated images."""
    from oracle import seg2eye_oracle as O
    from seg2eye_amd.options import parse
    from seg2eye_amd.pix2pix_model import Pix2PixModel
    from seg2eye_amd.tester import Tester
    import seg2eye_amd.data as data_mod
    store = _fake_openeds_store(seed=3)
    argv = ['--name', 'oe', '--checkpoints_dir', str(tmp_path), '--dataset_mode', 'openeds', '--dataset_key', 'validation', '--ngf', '8',
            '--crop_size', '256', '--aspect_ratio', '0.8', '--batchSize', '2', '--style_sample_method', 'first', '--compute_dtype', 'fp32']
    opt = parse(argv, is_train=False)
    orig = data_mod.create_dataloader
    data_mod.create_dataloader = lambda o, *a, **k: orig(o, store=store)
    try:
        tester = Tester(opt, dataset_key='validation')
    finally:
        data_mod.create_dataloader = orig
    assert tester.N == 9
    torch.manual_seed(0)
    Pix2PixModel(parse(argv)).save('latest')                 # a (randomly initialised) checkpoint for test mode to load
    model = Pix2PixModel(opt)
    model.eval()
    errs, stats = tester.run(model, mode='full')
    assert len(errs) == 9 and abs(stats['mse/validation/full/relative'] - float(np.mean(errs)) * 1471) < 1e-3
    errs_fix, _ = tester.run(model, mode='fix', limit=6)
    assert len(errs_fix) == 6
    # one batch against the oracle: same generated image -> resize + truncation + error on the CPU
    b = next(iter(tester.dataloader))
    e, fake, fake_resized, target = tester.run_batch(b, model)
    ref_resized = O.to_255_resized(fake.float().cpu())
    d = (fake_resized.cpu().int() - ref_resized).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-4             # (only within 1e-6 of a truncation boundary)
    ref_err = O.mse_for_images(fake_resized.cpu().int(), b['target_original'].int())
    np.testing.assert_allclose(e, ref_err.numpy(), rtol=1e-6)
    # the error log (util/tester.py:67-91): one row per sample with error / user / filename (npz here: no h5py in this image)
    errs_log, _ = tester.run(model, mode='full', write_error_log=True)
    import glob
    logs = glob.glob(os.path.join(tester.results_dir, 'error_log_validation.*'))
    assert len(logs) == 1, logs
    if logs[0].endswith('.npz'):
        zl = np.load(logs[0])
        np.testing.assert_allclose(zl['error'][:len(errs_log)], np.asarray(errs_log, dtype=np.float64))
        assert zl['user'].dtype == np.dtype('S4') and zl['filename'].dtype == np.dtype('S13') and zl['user'][0] != b''



def test_prepare_openeds_builds_the_store_the_dataset_reads(tmp_path):
    """SURVEY 8 f4 (data/prepare_openeds.py:16-138): a miniature OpenEDS folder tree -> the per-user store -> OpenEDSDataset."""
    from PIL import Image
    from seg2eye_amd.options import parse
    from seg2eye_amd.openeds_dataset import OpenEDSDataset
    from seg2eye_amd.prepare_openeds import OpenEDSPreparator, load_store
    rng = np.random.RandomState(0)
    base = str(tmp_path)
    P = OpenEDSPreparator
    users = {'train': ['U111', 'U112'], 'validation': ['U211'], 'test': ['U311']}
    truth = {}
    for subset, ids in users.items():
        mapping = []
        for u in ids:
            names = {k: ['%s%s%03d.png' % (u, k[0], i) for i in range(n)] for k, n in (('semantic_segmenation_images', 3), ('generative_images', 4), ('sequence_images', 2))}
            mapping.append({'id': u, **names})
            for key, folder in (('semantic_segmenation_images', os.path.join(P.FOLDER_SEMANTIC_SEGMENTATION, subset, 'images')),
                                ('generative_images', os.path.join(P.FOLDER_GENERATIVE, subset)),
                                ('sequence_images', os.path.join(P.FOLDER_SEQUENTIAL, subset))):
                if subset == 'test' and key == 'generative_images':
                    continue                                            # the test split has no generative images, only their labels
                os.makedirs(os.path.join(base, folder), exist_ok=True)
                for name in names[key]:
                    img = rng.randint(0, 256, size=(640, 400)).astype(np.uint8)
                    truth[name] = img
                    Image.fromarray(img, mode='L').save(os.path.join(base, folder, name))
            lab_src = names['generative_images'] if subset == 'test' else names['semantic_segmenation_images']
            lab_dir = os.path.join(base, P.FOLDER_GENERATIVE, subset, 'labels') if subset == 'test' else \
                os.path.join(base, P.FOLDER_SEMANTIC_SEGMENTATION, subset, 'labels')
            os.makedirs(lab_dir, exist_ok=True)
            for name in lab_src:
                lab = rng.randint(0, 4, size=(640, 400)).astype(np.uint8)
                truth['label:' + name] = lab
                np.save(os.path.join(lab_dir, name[:-3] + 'npy'), lab)
        with open(os.path.join(base, 'OpenEDS_%s_userID_mapping_to_images.json' % subset), 'w') as f:
            json.dump(mapping, f)
    out = OpenEDSPreparator(base, n_jobs=1, out_filename='mini.h5').run()
    store = load_store(out) if out.endswith('.npz') else None
    if store is None:
        import h5py
        store = h5py.File(out, 'r')
    g = store['train']['U112']
    assert g['images_ss'].shape == (3, 640, 400) and g['labels_ss'].shape == (3, 640, 400) and g['images_gen'].shape == (4, 640, 400)
    assert bytes(g['images_ss_filenames'][1]) == b'U112s001' and bytes(g['labels_ss_filenames'][1]) == b'U112s001.png'
    np.testing.assert_array_equal(np.asarray(g['images_gen'][2]), truth['U112g002.png'])
    np.testing.assert_array_equal(np.asarray(g['labels_ss'][0]), truth['label:U112s000.png'])
    assert 'labels_gen' in store['test']['U311'] and 'images_gen' not in store['test']['U311']
    # ... and the dataset walks it
    opt = parse(['--name', 'p', '--checkpoints_dir', str(tmp_path), '--dataset_mode', 'openeds', '--dataset_key', 'train', '--crop_size', '64',
                 '--aspect_ratio', '0.8', '--style_sample_method', 'first', '--no_flip'])
    ds = OpenEDSDataset(opt, store=store)
    assert len(ds) == 6
    it = ds[4]                                                          # second user, second sample
    assert it['user'] == 'U112' and it['label'].shape == (80, 64) and it['style_image'].shape == (4, 1, 80, 64)
