"""CPU-side checks: the C-ABI library loads and exports exactly what include/seg2eye_hip.h declares;
host logic (options, synthetic data, state-dict compatibility, flat Adam arenas, fail-loud on CPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, manifest_of


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'seg2eye_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(s2e_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from seg2eye_amd import _lib
    import __graft_entry__
    __graft_entry__.build()                       # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(_lib.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    L = _lib.lib()
    assert L.s2e_version() >= 1
    assert L.s2e_conv_cout_pad(1) == 32 and L.s2e_conv_cout_pad(64) == 64 and L.s2e_conv_cout_pad(200) == 256
    assert L.s2e_conv_k_pad(_lib.S2E_BF16, 9 * 128) == 1152 and L.s2e_conv_k_pad(_lib.S2E_F32, 80) == 96


def test_argument_errors_are_reported_without_a_gpu():
    from seg2eye_amd import _lib
    L = _lib.lib()
    assert L.s2e_colsum(_lib.S2E_BF16, None, 10, 8, None, None) == -1          # S2E_ERR_ARG before any launch
    assert b's2e_colsum' in L.s2e_last_error()
    d = _lib.ConvDesc(1, 8, 8, 8, 8, 8, 8, 3, 3, 3, 1, 0, 0, 0, 0)             # stride 3 unsupported
    assert L.s2e_conv2d(_lib.S2E_BF16, 1, 1, None, None, None, 1, ctypes.byref(d), None, 0, None) == -3
    d2 = _lib.ConvDesc(8, 8, 8, 1024, 8, 8, 1024, 3, 3, 1, 1, 0, 0, 0, 0)         # small M, K = 9216: split-K
    assert L.s2e_conv2d_workspace_bytes(_lib.S2E_BF16, ctypes.byref(d2)) == 16 * 512 * 1024 * 4
    assert L.s2e_conv2d(_lib.S2E_BF16, 1, 1, None, None, None, 1, ctypes.byref(d2), None, 0, None) == -1   # workspace missing
    d3 = _lib.ConvDesc(8, 256, 256, 128, 256, 256, 256, 3, 3, 1, 1, 0, 0, 0, 0)
    assert L.s2e_conv2d_workspace_bytes(_lib.S2E_BF16, ctypes.byref(d3)) == 0
    # big bf16 3x3 layer: patch-resident weight gradient, one workgroup per CU (256 when no device is visible), each
    # storing its 9 x 128 x 64 fp32 partial tile; the fp32 build of the same shape runs the generic kernel, whose 2 x 9 dW tiles
    # x 114 pixel splits (>= 16 splits: partial tiles + a fixed-order reduction instead of atomics, round 3) each store a
    # 128 x 128 tile and 128 bias sums
    assert L.s2e_conv2d_wgrad_workspace_bytes(_lib.S2E_BF16, ctypes.byref(d3)) == 256 * 9 * 128 * 64 * 4
    assert L.s2e_conv2d_wgrad_workspace_bytes(_lib.S2E_F32, ctypes.byref(d3)) == 18 * 114 * (128 * 128 + 128) * 4
    # which kernel the library picks is a function of the shape alone (0 generic, 1 one-channel stream, 2 patch-resident)
    assert L.s2e_conv2d_kernel_kind(_lib.S2E_BF16, ctypes.byref(d3)) == 2
    assert L.s2e_conv2d_kernel_kind(_lib.S2E_F32, ctypes.byref(d3)) == 2
    assert L.s2e_conv2d_wgrad_kernel_kind(_lib.S2E_BF16, ctypes.byref(d3)) == 2
    assert L.s2e_conv2d_wgrad_kernel_kind(_lib.S2E_F32, ctypes.byref(d3)) == 0
    assert L.s2e_conv2d_kernel_kind(_lib.S2E_BF16, ctypes.byref(d2)) == 0                      # 8x8: rectangles would be 25 % full
    d4 = _lib.ConvDesc(8, 256, 256, 64, 256, 256, 1, 3, 3, 1, 1, 0, 0, 0, 0)
    assert L.s2e_conv2d_kernel_kind(_lib.S2E_BF16, ctypes.byref(d4)) == 1                      # conv_img: Cout = 1
    d5 = _lib.ConvDesc(8, 16, 16, 1024, 16, 16, 1024, 3, 3, 1, 1, 0, 0, 0, 0)                  # few tiles, long K: patch kernel, 64-channel tiles split 2x
    assert L.s2e_conv2d_kernel_kind(_lib.S2E_BF16, ctypes.byref(d5)) == 2                      # (round 5; 128-channel tiles split 4x before: twice the fp32 slabs)
    assert L.s2e_conv2d_workspace_bytes(_lib.S2E_BF16, ctypes.byref(d5)) == 2 * 8 * 16 * 16 * 1024 * 4
    d4 = _lib.ConvDesc(8, 256, 256, 64, 256, 256, 1, 3, 3, 1, 1, 0, 0, 0, 0)      # conv_img: 1-channel stream kernels
    assert L.s2e_conv2d_wgrad_workspace_bytes(_lib.S2E_BF16, ctypes.byref(d4)) == 1024 * 9 * 64 * 4
    assert L.s2e_conv2d_wgrad(_lib.S2E_BF16, 1, 1, 1, None, ctypes.byref(d4), None, 0, None) == -1        # workspace missing
    with pytest.raises(_lib.Seg2EyeHipError):
        _lib.check(-1, 'x')


def test_round6_planners_are_host_only():
    """The shape planners of round 6's kernels answer without a GPU: which launches conv_plane.hip takes (and in which mode), the
    size of a PLANE-layout weight, which weight gradients the multi-job launch takes; argument errors before any launch."""
    from seg2eye_amd import _lib
    L = _lib.lib()
    bf, f32 = _lib.S2E_BF16, _lib.S2E_F32
    mode = lambda *a: L.s2e_conv2d_plane_supported(bf, ctypes.byref(_lib.ConvDesc(*a)))
    # netE at the benchmark's size: batch 32, 3x3 stride 2 pad 1 (forward = mode 3, data gradient = mode 4)
    assert mode(32, 128, 128, 64, 64, 64, 128, 3, 3, 2, 1, 0, 0, 0, 0) == 3
    assert mode(32, 64, 64, 128, 128, 128, 64, 3, 3, 2, 1, 1, 0, 0, 0) == 4
    assert mode(32, 16, 16, 512, 8, 8, 512, 3, 3, 2, 1, 0, 0, 0, 0) == 0                # an 8 x 8 map: no 8 x 16 rectangle
    # the learned 1x1 shortcuts (mode 1), forward and data gradient
    assert mode(8, 256, 256, 128, 256, 256, 64, 1, 1, 1, 0, 0, 0, 0, 0) == 1
    assert mode(8, 256, 256, 64, 256, 256, 128, 1, 1, 1, 0, 1, 0, 0, 0) == 1
    # the PatchGAN's 4x4 stride-2 pad-2 layers on ragged maps (modes 5 / 6); its 8-channel first layer stays generic
    assert mode(16, 129, 129, 64, 65, 65, 128, 4, 4, 2, 2, 0, 0, 0, 0) == 5
    assert mode(8, 65, 65, 128, 129, 129, 64, 4, 4, 2, 2, 1, 0, 0, 0) == 6
    assert mode(16, 256, 256, 8, 129, 129, 64, 4, 4, 2, 2, 0, 0, 1, 0) == 0
    # 3x3 stride 1 stays in conv_duo.hip by default; fp32 never runs here
    assert mode(8, 128, 128, 256, 128, 128, 128, 3, 3, 1, 1, 0, 0, 0, 0) == 0
    assert L.s2e_conv2d_plane_supported(f32, ctypes.byref(_lib.ConvDesc(8, 256, 256, 128, 256, 256, 64, 1, 1, 1, 0, 0, 0, 0, 0))) == 0
    d = _lib.ConvDesc(32, 128, 128, 64, 64, 64, 200, 3, 3, 2, 1, 0, 0, 0, 0)
    assert L.s2e_conv_plane_weight_elems(ctypes.byref(d)) == 256 * 9 * 64                # rows padded to 64
    assert L.s2e_conv2d_plane(bf, 1, 1, None, None, None, 1, ctypes.byref(_lib.ConvDesc(8, 8, 8, 64, 8, 8, 64, 3, 3, 1, 1, 0, 0, 0, 0)), None) == -3
    assert b's2e_conv2d_plane' in L.s2e_last_error()
    assert L.s2e_conv2d_plane(bf, None, 1, None, None, None, 1, ctypes.byref(d), None) == -1
    # multi-job generic weight gradient: generic bf16 shapes only
    multi = lambda *a: L.s2e_conv2d_wgrad_multi_supported(bf, ctypes.byref(_lib.ConvDesc(*a)))
    assert multi(32, 128, 128, 64, 64, 64, 128, 3, 3, 2, 1, 0, 0, 0, 0) == 1             # netE stride 2
    assert multi(8, 256, 256, 128, 256, 256, 64, 1, 1, 1, 0, 0, 0, 0, 0) == 1            # 1x1 shortcut
    assert multi(8, 256, 256, 128, 256, 256, 256, 3, 3, 1, 1, 0, 0, 0, 0) == 0           # patch-resident: the batched launch's
    assert multi(8, 256, 256, 64, 256, 256, 1, 3, 3, 1, 1, 0, 0, 0, 0) == 0              # 1-channel stream kernel's
    assert L.s2e_conv2d_wgrad_multi(bf, None, 0, None, 0, None) == -1
    # ... and which of them leave the generic tile kernel for the flat-slab patch-resident one (default: the PatchGAN's 4x4 layers)
    if 'S2E_WGRAD_FLAT' not in os.environ:
        kind = lambda *a: L.s2e_conv2d_wgrad_multi_kind(bf, ctypes.byref(_lib.ConvDesc(*a)))
        assert kind(16, 129, 129, 64, 65, 65, 128, 4, 4, 2, 2, 0, 0, 0, 0) == 5          # 4x4 stride 2 on a ragged map
        assert kind(16, 33, 33, 256, 34, 34, 512, 4, 4, 1, 2, 0, 0, 0, 0) == 4           # 4x4 stride 1
        assert kind(32, 128, 128, 64, 64, 64, 128, 3, 3, 2, 1, 0, 0, 0, 0) == 0          # netE: stays generic by default
        if 'S2E_CONV_C8' not in os.environ:
            assert kind(16, 256, 256, 8, 129, 129, 64, 4, 4, 2, 2, 0, 0, 0, 0) == 6      # the PatchGAN's first layer: conv_c8.hip
            assert L.s2e_conv2d_kernel_kind(bf, ctypes.byref(_lib.ConvDesc(16, 256, 256, 8, 129, 129, 64, 4, 4, 2, 2, 0, 0, 1, 0))) == 1   # S2E_KERNEL_SMALL
        assert kind(16, 256, 256, 16, 129, 129, 64, 4, 4, 2, 2, 0, 0, 0, 0) == 0         # 16 input channels: generic
        assert kind(16, 33, 33, 256, 34, 34, 512, 4, 4, 1, 2, 0, 1, 0, 0) == 0           # an input activation: generic
        assert kind(8, 256, 256, 128, 256, 256, 256, 3, 3, 1, 1, 0, 0, 0, 0) == -1       # not a job of the multi call at all
    jobs = (_lib.WgradMultiJob * 2)()
    for j, a in zip(jobs, ((32, 128, 128, 64, 64, 64, 128, 3, 3, 2, 1, 0, 0, 0, 0), (8, 256, 256, 128, 256, 256, 64, 1, 1, 1, 0, 0, 0, 0, 0))):
        j.d = _lib.ConvDesc(*a)
    assert L.s2e_conv2d_wgrad_multi_workspace_bytes(bf, ctypes.byref(jobs), 2) % 256 == 0
    assert L.s2e_conv2d_wgrad_multi(bf, ctypes.byref(jobs), 2, None, 0, None) == -1      # null operand pointers
    assert L.s2e_shard_sum(bf, 1, 1, 8, 100, None) == -1                                  # shard bytes not a multiple of 16


def test_pack_block_map_is_host_only():
    """s2e_pack_block_map is pure host code: grid of a forward pack = rows_pad/4 x ceil(cin_pad/64), of a
    transposed pack = ceil(cout/64) x ceil(rows_pad/8); triples are {job, bx, by}."""
    from seg2eye_amd import _lib
    L = _lib.lib()
    jobs = (_lib.PackJob * 2)()
    jobs[0].cout, jobs[0].cin, jobs[0].taps, jobs[0].cin_pad, jobs[0].transposed = 100, 70, 9, 72, 0
    jobs[1].cout, jobs[1].cin, jobs[1].taps, jobs[1].cin_pad, jobs[1].transposed = 100, 70, 9, 72, 1
    n0 = L.s2e_pack_block_map(_lib.S2E_BF16, ctypes.byref(jobs), 1, None)
    n = L.s2e_pack_block_map(_lib.S2E_BF16, ctypes.byref(jobs), 2, None)
    assert n0 == 128 // 4 * 2 and n == n0 + 2 * (128 // 8)
    bm = np.zeros(3 * n, dtype=np.int32)
    assert L.s2e_pack_block_map(_lib.S2E_BF16, ctypes.byref(jobs), 2, bm.ctypes.data) == n
    bm = bm.reshape(-1, 3)
    assert (bm[:n0, 0] == 0).all() and (bm[n0:, 0] == 1).all()
    assert bm[:n0, 1].max() == 31 and bm[:n0, 2].max() == 1 and bm[n0:, 1].max() == 1 and bm[n0:, 2].max() == 15
    assert len({tuple(r) for r in bm}) == n
    assert L.s2e_pack_conv_weights(_lib.S2E_BF16, None, None, 0, 9, None, None) == -1


def test_grad_block_map_is_host_only():
    """s2e_grad_block_map: tiles of 4 co rows x 64 ci, up to 8 tiles per block, triples {job, first tile, tiles}."""
    from seg2eye_amd import _lib
    L = _lib.lib()
    jobs = (_lib.GradJob * 2)()
    jobs[0].cout, jobs[0].cin, jobs[0].taps, jobs[0].cin_pad = 10, 130, 9, 136      # 3 row groups x 3 chunks = 9 tiles
    jobs[1].cout, jobs[1].cin, jobs[1].taps, jobs[1].cin_pad = 4, 64, 1, 64         # 1 tile
    n = L.s2e_grad_block_map(ctypes.byref(jobs), 2, None)
    assert n == 2 + 1
    bm = np.zeros(3 * n, dtype=np.int32)
    assert L.s2e_grad_block_map(ctypes.byref(jobs), 2, bm.ctypes.data) == n
    assert bm.reshape(-1, 3).tolist() == [[0, 0, 8], [0, 8, 1], [1, 0, 1]]
    assert L.s2e_weight_grads_batched(None, None, 0, 9, 0, None, None) == -1


def test_ops_refuse_cpu_tensors():
    from seg2eye_amd import ops, _lib, networks
    from seg2eye_amd.options import default_opt
    with pytest.raises(_lib.Seg2EyeHipError):
        ops.in_stats(torch.zeros(1, 4, 4, 8))
    G = networks.SPADESTYLEGenerator(default_opt(ngf=8, crop_size=64, gpu_ids=[]))
    with pytest.raises(_lib.Seg2EyeHipError):
        G(torch.zeros(1, 4, 64, 64), torch.zeros(1, 16))


def test_state_dict_layout_matches_reference():
    from seg2eye_amd import networks
    from seg2eye_amd.options import default_opt
    z = load_golden('trainer_ngf8_256')
    opt = default_opt(ngf=8, ndf=8, crop_size=256, gpu_ids=[])
    for tag, cls in (('G', networks.SPADESTYLEGenerator), ('D', networks.MultiscaleDiscriminator), ('E', networks.ConvEncoder)):
        sd = cls(opt).state_dict()
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == manifest_of(z, tag)
    big = default_opt(ngf=64, ndf=64, crop_size=256, gpu_ids=[])
    n = lambda net: sum(p.numel() for p in net.parameters())
    assert abs(n(networks.SPADESTYLEGenerator(big)) / 1e6 - 92.46) < 0.01
    assert abs(n(networks.MultiscaleDiscriminator(big)) / 1e6 - 5.53) < 0.01
    assert abs(n(networks.ConvEncoder(big)) / 1e6 - 6.53) < 0.01


def test_options_and_shapes():
    from seg2eye_amd.options import default_opt, latent_size, image_hw
    assert latent_size(default_opt(crop_size=256, aspect_ratio=1.0)) == (8, 8)
    assert image_hw(default_opt(crop_size=256, aspect_ratio=0.8)) == (320, 256)
    assert image_hw(default_opt(crop_size=384, aspect_ratio=0.6)) == (640, 384)
    with pytest.raises(ValueError):
        latent_size(default_opt(num_upsampling_layers='most'))
    with pytest.raises(KeyError):
        default_opt(not_an_option=1)
    from seg2eye_amd.pix2pix_model import Pix2PixModel
    bn = Pix2PixModel(default_opt(ngf=8, ndf=8, gpu_ids=[], norm_G='spectralspadebatch3x3'))     # the reference's default norm
    assert len(bn.netG.state_dict()) == 270 and 'head_0.norm_0.spade.param_free_norm.running_mean' in bn.netG.state_dict()
    with pytest.raises(ValueError):
        Pix2PixModel(default_opt(ngf=8, ndf=8, gpu_ids=[], norm_G='spectralspadesyncbatch3x3'))
    Pix2PixModel(default_opt(ngf=8, ndf=8, gpu_ids=[], lambda_openeds=1.0))          # built since SURVEY 8 f3 (s2e_openeds_error)
    with pytest.raises(NotImplementedError):
        Pix2PixModel(default_opt(ngf=8, ndf=8, gpu_ids=[], no_vgg_loss=False))


def test_synthetic_is_deterministic_and_wellformed():
    from seg2eye_amd import synthetic as syn
    a, b = syn.make_batch(2, 64, 64, seed=5), syn.make_batch(2, 64, 64, seed=5)
    assert all(np.array_equal(a[k], b[k]) for k in ('label', 'style_image', 'target'))
    assert a['label'].dtype == np.uint8 and a['label'].shape == (2, 1, 64, 64) and set(np.unique(a['label'])) <= {0, 1, 2, 3}
    assert len(np.unique(a['label'])) == 4
    assert a['style_image'].shape == (2, 4, 1, 64, 64) and np.abs(a['style_image']).max() <= 1.0
    assert not np.array_equal(a['label'], syn.make_batch(2, 64, 64, seed=6)['label'])
    sd = syn.fill_state_dict([('c.weight_orig', (8, 4, 3, 3)), ('c.weight_u', (8,)), ('c.weight_v', (36,)), ('c.bias', (8,))])
    wm = sd['c.weight_orig'].reshape(8, -1).astype(np.float64)
    sigma = sd['c.weight_u'] @ wm @ sd['c.weight_v']
    assert abs(sigma - np.linalg.svd(wm, compute_uv=False)[0]) < 1e-3 * sigma       # settled to the top singular pair


def test_flat_adam_arena_aliases_parameters():
    from seg2eye_amd.optim import FlatAdam
    lin = torch.nn.Linear(5, 3)
    w0 = lin.weight.detach().clone()
    opt = FlatAdam(list(lin.parameters()), lr=1e-3, betas=(0, 0.9))
    assert opt.betas == (0.0, 0.9) and isinstance(opt.betas[0], float)
    assert torch.equal(lin.weight.detach(), w0)
    assert lin.weight.data_ptr() == opt.flat_p.data_ptr() and lin.weight.grad.data_ptr() == opt.flat_g.data_ptr()
    assert all(o % 4 == 0 for o in opt.offsets) and opt.numel == 16 + 4
    lin(torch.ones(2, 5)).sum().backward()
    assert float(opt.flat_g.abs().sum()) > 0 and lin.weight.grad.data_ptr() == opt.flat_g.data_ptr()
    opt.zero_grad()
    assert float(opt.flat_g.abs().sum()) == 0
    with pytest.raises(Exception):
        opt.step()                                                                        # GPU-only kernel
    assert float(FlatAdam(list(torch.nn.Linear(2, 2).parameters()), lr=1e-3, weight_decay=0.1).hyper[6]) == pytest.approx(0.1)


def test_flat_adam_channels_last_arena_layout():
    """Conv weights with Cin % 8 == 0 are stored [Cout][KH][KW][Cin] in the arenas, 256-byte aligned, and exposed as
    (Cout,Cin,KH,KW) views: same values, state_dict unchanged; the moments' layout is tagged and a mismatching load refused."""
    from seg2eye_amd.optim import FlatAdam
    from seg2eye_amd import optim as optim_mod
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.Conv2d(16, 24, 3), torch.nn.Conv2d(4, 8, 3), torch.nn.Linear(5, 3))
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    opt = FlatAdam(list(net.parameters()), lr=1e-3, channels_last=True)
    w0, w2 = net[0].weight, net[2].weight
    assert opt.cl == [True, False, True, False, False, False, False, False]            # (Cin = 4 keeps torch's order)
    assert w0.shape == (16, 8, 3, 3) and w0.stride() == (72, 1, 24, 8) and not w0.is_contiguous()
    assert w0.permute(0, 2, 3, 1).is_contiguous() and w0.grad.stride() == w0.stride() and w2.is_contiguous()
    assert all(opt.offsets[i] % 64 == 0 for i, c in enumerate(opt.cl) if c)
    assert (w0.data_ptr() - opt.flat_p.data_ptr()) % 256 == 0
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k                                              # values by logical index: unchanged
    # the arena really is [co][ky][kx][ci]
    off = opt.offsets[0]
    assert torch.equal(opt.flat_p[off:off + w0.numel()].view(16, 3, 3, 8), before['0.weight'].permute(0, 2, 3, 1))
    x = torch.randn(2, 8, 12, 12)
    net[:2](x).sum().backward()                                                         # torch's own kernels accumulate into the views
    ref = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.Conv2d(16, 24, 3))
    ref.load_state_dict({k: v for k, v in before.items() if k[0] in '01'})
    ref(x).sum().backward()
    assert torch.allclose(w0.grad, ref[0].weight.grad, atol=1e-5) and w0.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off
    sd = opt.state_dict()
    assert sd['layout']['channels_last'] == opt.cl
    opt.load_state_dict(sd)
    #  a state carries its parameter signature: the same parameters in another MEMORY order are converted by logical
    # index, a permuted parameter list is refused, a state without any signature is refused unless the caller vouches for its order
    opt.flat_m.copy_(torch.arange(opt.numel, dtype=torch.float32))
    opt.flat_v.copy_(torch.arange(opt.numel, dtype=torch.float32) * 0.5)
    opt.step_count = 7
    sd = opt.state_dict()
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3), torch.nn.Conv2d(16, 24, 3), torch.nn.Conv2d(4, 8, 3), torch.nn.Linear(5, 3))
    other = FlatAdam(list(mk().parameters()), lr=1e-3, channels_last=False)
    other.load_state_dict(sd)
    assert other.step_count == 7
    def same_moments(a, b):
        for i, (pa, pb) in enumerate(zip(a.params, b.params)):
            for fa, fb in ((a.flat_m, b.flat_m), (a.flat_v, b.flat_v)):
                if not torch.equal(optim_mod._arena_view(fa, a.offsets[i], pa, a.cl[i]), optim_mod._arena_view(fb, b.offsets[i], pb, b.cl[i])):
                    return False                                                          # (same moment for the same logical weight element)
        return True
    assert same_moments(opt, other)
    back = FlatAdam(list(mk().parameters()), lr=1e-3, channels_last=True)
    back.load_state_dict(other.state_dict())                                              # torch order -> channels-last
    assert same_moments(opt, back)
    prm = list(mk().parameters())
    permuted = FlatAdam([prm[2], prm[3], prm[0], prm[1]] + prm[4:], lr=1e-3, channels_last=False)
    with pytest.raises(ValueError, match='another parameter list'):
        permuted.load_state_dict(other.state_dict())
    legacy = {k: v for k, v in other.state_dict().items() if k != 'layout'}               # what S2E_WEIGHTS_CL=0 builds before round 5 wrote
    with pytest.raises(ValueError, match='no layout'):
        back.load_state_dict(legacy)
    back.flat_m.zero_()
    back.load_state_dict(legacy, trust_param_order=True)
    assert same_moments(opt, back)
    unsigned = dict(sd, layout={k: v for k, v in sd['layout'].items() if k != 'shapes'})  # round 3/4 states: layout without signature
    opt.load_state_dict(unsigned)
    with pytest.raises(ValueError, match='no parameter signature'):
        other.load_state_dict(unsigned)


def test_channels_last_pack_and_inplace_gradient_maps_are_host_only():
    """s2e_pack_block_map with the channels-last source bit (transposed | 2) and s2e_sngrad_block_map / _scratch_floats: pure
    host code."""
    from seg2eye_amd import _lib
    L = _lib.lib()
    jobs = (_lib.PackJob * 2)()
    jobs[0].cout, jobs[0].cin, jobs[0].taps, jobs[0].cin_pad, jobs[0].transposed = 256, 128, 9, 128, 2      # forward: 2048 elements per block
    jobs[1].cout, jobs[1].cin, jobs[1].taps, jobs[1].cin_pad, jobs[1].transposed = 256, 128, 9, 128, 3      # transposed: 64 x 64 tile per tap
    n0 = L.s2e_pack_block_map(_lib.S2E_BF16, ctypes.byref(jobs), 1, None)
    n = L.s2e_pack_block_map(_lib.S2E_BF16, ctypes.byref(jobs), 2, None)
    assert n0 == 256 * 1152 // 2048 and n - n0 == (256 // 64) * 9 * (128 // 64)
    sj = (_lib.SnGradJob * 2)()
    sj[0].rows, sj[0].cin, sj[0].taps = 64, 64, 9          # 36864 elements: 3 chunks of 16384
    sj[1].rows, sj[1].cin, sj[1].taps = 8, 8, 1            # 64 elements: 1 chunk
    nb = L.s2e_sngrad_block_map(ctypes.byref(sj), 2, None)
    assert nb == 4 and (sj[0].part0, sj[0].nparts, sj[1].part0, sj[1].nparts) == (0, 3, 3, 1)
    bm = np.zeros(2 * nb, dtype=np.int32)
    assert L.s2e_sngrad_block_map(ctypes.byref(sj), 2, bm.ctypes.data) == nb
    assert bm.reshape(-1, 2).tolist() == [[0, 0], [0, 1], [0, 2], [1, 0]]
    assert (sj[0].vmem0, sj[1].vmem0) == (4, 4 + 576) and L.s2e_sngrad_scratch_floats(ctypes.byref(sj), 2) == 4 + 576 + 8
    assert L.s2e_sn_grads_inplace(None, None, 0, None, None) == -1


def test_checkpoint_roundtrip_and_module_prefix(tmp_path):
    from seg2eye_amd import checkpoint, networks
    from seg2eye_amd.options import default_opt
    opt = default_opt(ngf=8, ndf=8, crop_size=64, gpu_ids=[], checkpoints_dir=str(tmp_path), name='t')
    D = networks.MultiscaleDiscriminator(opt)
    path = checkpoint.save_network(D, 'D', 'latest', opt)
    assert path.endswith('latest_net_D.pth')                                              # util/util.py:195-200
    sd = torch.load(path)
    torch.save({'module.' + k: v for k, v in sd.items()}, path)                           # DataParallel-style keys
    D2 = networks.MultiscaleDiscriminator(opt)
    checkpoint.load_network(D2, 'D', 'latest', opt)
    for (k, a), (_, b) in zip(D.state_dict().items(), D2.state_dict().items()):
        assert torch.equal(a, b), k
