"""Randomised 3x3 / 4x4 stride-1 conv shapes (forward, data-gradient, weight/bias/residual gradients) against an fp64 reference on the
GPU.  Run by test_ops_gpu.py::test_conv2d_random_shapes in a child process so that the kernel-selection thresholds
(S2E_CONV_PATCH / S2E_WGRAD_PATCH, read once at library load) can be lowered: every eligible shape then takes the
patch-resident kernels, whatever its size.  Not collected by pytest (leading underscore)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from seg2eye_amd import ops
dev = torch.device('cuda:0')
random.seed(int(os.environ.get('SEED', '1')))
torch.manual_seed(int(os.environ.get('SEED', '1')))
bad = 0
for it in range(int(os.environ.get('N', '24'))):
    dt = random.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    N = random.choice([1, 2, 3, 5]); H = random.choice([16, 17, 31, 32, 48, 64, 70, 96]); W = random.choice([16, 24, 32, 33, 40, 64, 80, 100, 128])
    cin = random.choice([64, 128, 192, 256]); cout = random.choice([40, 64, 72, 128, 136, 256])
    if random.random() < 0.25: cin = 8; cout = random.choice([128, 256])
    has_b = random.random() < 0.7; has_r = random.random() < 0.4; out_act = random.choice([0, 0, 1, 2])
    k = 3 if (cin == 8 or random.random() < 0.7) else 4            # 4x4: the PatchGAN's stride-1 layers (pad 2: output grows by 1)
    pad = 1 if k == 3 else 2
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    x = torch.randn(N, H, W, cin, device=dev).to(dt).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).requires_grad_(True)
    b = (0.1 * torch.randn(cout, device=dev)).requires_grad_(True) if has_b else None
    r = torch.randn(N, Ho, Wo, cout, device=dev).to(dt).requires_grad_(True) if has_r else None
    gy = torch.randn(N, Ho, Wo, cout, device=dev).to(dt)
    # fp64 reference on the GPU
    xr = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True); wr = w.detach().to(dt).double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True) if has_b else None
    rr = r.detach().double().permute(0, 3, 1, 2).requires_grad_(True) if has_r else None
    yr = F.conv2d(xr, wr, br, padding=pad)
    if has_r: yr = yr + rr
    if out_act == 1:
        # LeakyReLU kink: a pre-activation within rounding of 0 may fall on either side in two implementations (and then
        # the whole 0.8 * gy of that element differs); such elements carry no incoming gradient in this test
        gy = gy * (yr.detach().abs() > 1e-3).permute(0, 2, 3, 1).to(gy.dtype)
        yr = F.leaky_relu(yr, 0.2)
    if out_act == 2: yr = torch.tanh(yr)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    y = ops.conv2d(x, w, b, r, 1, pad, 0, out_act)
    y.backward(gy)
    tol = 2e-2 if dt == torch.bfloat16 else 2e-4
    def chk(name, got, ref):
        global bad
        got, ref = got.detach(), ref.detach()
        s = max(float(ref.abs().max()), 1e-6); e = float((got.double() - ref).abs().max())
        if not e <= tol * s:
            bad += 1; print('MISMATCH', name, 'err %.3e scale %.3e' % (e, s), (N, H, W, cin, cout, k, dt, has_b, has_r, out_act))
    chk('y', y.permute(0, 3, 1, 2), yr.detach()); chk('dx', x.grad.permute(0, 3, 1, 2), xr.grad); chk('dw', w.grad, wr.grad)
    if has_b: chk('db', b.grad, br.grad)
    if has_r: chk('dres', r.grad.permute(0, 3, 1, 2), rr.grad)
print('done, mismatches:', bad)
