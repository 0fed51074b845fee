#!/usr/bin/env python3
"""s2e_wgrad_batch (csrc/conv_wgrad_batch.hip: the queued 3x3 weight gradients of a backward pass as ONE stream-K launch) against
an fp64 evaluation of the same sums on the same bf16 operands, and against the per-layer launches it replaces.

    python tools/check_wgrad_batch.py [--bench]         (S2E_WGRAD_BATCH_WGS=<n> changes the number of workgroups)

Cases: tiles with a single owner (added straight into dW) and tiles shared between workgroups (fragments + fix-up), Cout below
and not a multiple of 128, several ci tiles, jobs with and without bias, label-sparse jobs (device-side rectangle lists: empty,
odd, full), accumulation into a non-zero dW, more jobs than one launch holds.  tests/test_ops_gpu.py runs it with the default
workgroup count (one per CU: nearly every tile of the small cases is shared) and with 5 workgroups (long ranges: whole tiles)."""
import argparse
import ctypes as C
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import _lib as L, ops      # noqa: E402


def ref_wgrad(x, gy, mask=None):
    """fp64: dw (Cout, 9*Cin) in (tap, ci) order and db (Cout) of a 3x3 stride-1 pad-1 conv; x, gy NHWC bf16."""
    xd = x.double().permute(0, 3, 1, 2)
    gd = gy.double().permute(0, 3, 1, 2)
    if mask is not None:
        gd = gd * mask[:, None].double()
    n, cin, h, w = xd.shape
    cout = gd.shape[1]
    wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    (F.conv2d(xd, wz, padding=1) * gd).sum().backward()
    return wz.grad.permute(0, 2, 3, 1).reshape(cout, 9 * cin), gd.sum(dim=(0, 2, 3))


def rect_mask(n, h, w, rects, dev):
    m = torch.zeros(n, h, w, device=dev)
    tx, ty = w // 16, h // 16
    for r in rects:
        m[r // (tx * ty), ((r // tx) % ty) * 16:((r // tx) % ty) * 16 + 16, (r % tx) * 16:(r % tx) * 16 + 16] = 1.0
    return m


def run_batch(jobs, dev):
    arr = (L.WgradBatchJob * len(jobs))()
    for a, j in zip(arr, jobs):
        x, gy, dw, db, rl, rc = j[:6]
        a.flags = j[6] if len(j) > 6 else 0
        a.x, a.gy, a.dw, a.dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
        a.rect_list, a.rect_count = (rl.data_ptr(), rc.data_ptr()) if rl is not None else (None, None)
        a.N, a.H, a.W, a.Cin = x.shape
        a.Cout = gy.shape[-1]
    wsb = L.lib().s2e_wgrad_batch_workspace_bytes()
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    ws.fill_(float('nan'))                               # a fragment read before it is written shows
    L.check(L.lib().s2e_wgrad_batch(L.S2E_BF16, C.byref(arr), len(jobs), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream), 's2e_wgrad_batch')
    return ws


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bench', action='store_true', help='also time the G step\'s job mix against the per-layer launches')
    ap.add_argument('--mix', default='all', choices=['all', 'long', 'short'], help='--bench: all 28 layers, the 256^2 / 128^2 ones, or the rest')
    ap.add_argument('--only-batched', action='store_true', help='--bench: time the batched launch only (for profiler runs)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(3)
    L.lib()
    g = torch.Generator(device='cpu').manual_seed(5)
    # (N, H, W, Cin, Cout, bias, rects: None | 'some' | 'none' | 'all')
    cases = [(2, 16, 16, 128, 256, True, None), (1, 32, 32, 64, 64, True, None), (2, 16, 32, 128, 136, False, None),
             (1, 64, 64, 64, 128, True, None), (2, 8, 16, 192, 72, True, None), (2, 32, 32, 64, 128, True, 'some'),
             (1, 32, 48, 128, 256, True, 'none'), (2, 16, 16, 64, 200, False, 'all'), (1, 8, 16, 512, 512, True, None),
             (3, 16, 16, 256, 64, True, 'some')]
    cases = cases + [(1, 8, 16, 64, 64 + 8 * i, bool(i & 1), None) for i in range(26)]       # > 32 jobs: two launches
    jobs, refs = [], []
    for i, (n, h, w, cin, cout, bias, rk) in enumerate(cases):
        assert L.lib().s2e_wgrad_batch_supported(L.S2E_BF16, n, h, w, cin, cout)
        x = torch.randn(n, h, w, cin, generator=g).to(dev).to(torch.bfloat16)
        gy = torch.randn(n, h, w, cout, generator=g).to(dev).to(torch.bfloat16)
        dw0 = torch.randn(cout, 9 * cin, generator=g).to(dev)            # accumulated INTO: starts non-zero
        db0 = torch.randn(cout, generator=g).to(dev) if bias else None
        rl = rc = mask = None
        if rk is not None:
            nr = n * (h // 16) * (w // 16)
            pick = {'some': [r for r in range(nr) if (r * 7 + 3) % 5 < 3][: max(1, nr - 1) | 1], 'none': [], 'all': list(range(nr))}[rk]
            rl = torch.full((nr + 3,), -12345, dtype=torch.int32, device=dev)       # (entries past the count must never be read as rectangles)
            if pick:
                rl[:len(pick)] = torch.tensor(pick, dtype=torch.int32)
            rc = torch.tensor([len(pick), 777], dtype=torch.int32, device=dev)
            mask = rect_mask(n, h, w, pick, dev)
        rw, rb = ref_wgrad(x, gy, mask)
        refs.append((dw0.double() + rw, (db0.double() + rb) if bias else None, float(rw.abs().max()), float(rb.abs().max())))
        jobs.append((x, gy, dw0.clone(), db0.clone() if bias else None, rl, rc))
    # S2E_WGRAD_BATCH_DW_ZERO: the same first cases into all-zero dW with the flag (single-owner tiles are stored, not added)
    nz = 10
    for i in range(nz):
        x, gy, dw0, db0, rl, rc = jobs[i]
        jobs.append((x, gy, torch.zeros_like(dw0), db0.clone() if db0 is not None else None, rl, rc, 1))
        rw, rb, sw, sb = refs[i]
        refs.append((rw - dw0.double(), rb, sw, sb))
        cases.append(cases[i])
    run_batch(jobs, dev)
    torch.cuda.synchronize()
    worst = 0.0
    for i, (j, (rw, rb, sw, sb)) in enumerate(zip(jobs, refs)):
        ew = float((j[2].double() - rw).abs().max()) / max(sw, 1e-6)
        worst = max(worst, ew)
        assert ew < 2e-5, ('dW of case %d %s' % (i, cases[i]), ew)       # fp32 accumulation of exact bf16 products
        if rb is not None:
            eb = float((j[3].double() - rb).abs().max()) / max(sb, 1e-6)
            worst = max(worst, eb)
            assert eb < 2e-5, ('dbias of case %d %s' % (i, cases[i]), eb)
    # two jobs into one dW are refused (single-owner tiles are added without atomics)
    twice = [jobs[0], (jobs[0][0], jobs[0][1], jobs[0][2], None, None, None)]
    try:
        run_batch(twice, dev)
        raise AssertionError('two jobs sharing dW must be refused')
    except L.Seg2EyeHipError as e:
        assert 'same dW' in str(e)
    print('worst relative error %.2e over %d jobs (workgroups: %s)' % (worst, len(jobs), os.environ.get('S2E_WGRAD_BATCH_WGS', 'one per CU')))

    if args.bench:
        # the patch-resident weight gradients of one G step at the bench's size (batch 8, ngf 64), dense
        mix = [(256, 128, 256), (256, 128, 256), (256, 128, 128), (256, 128, 64), (256, 64, 64),
               (128, 128, 512), (128, 128, 512), (128, 128, 256), (128, 256, 128), (128, 128, 128),
               (64, 128, 1024), (64, 128, 1024), (64, 128, 512), (64, 512, 256), (64, 256, 256),
               (32, 128, 2048), (32, 128, 2048), (32, 128, 1024), (32, 1024, 512), (32, 512, 512),
               (16, 128, 2048), (16, 128, 2048), (16, 128, 2048), (16, 128, 2048),
               (16, 1024, 1024), (16, 1024, 1024), (16, 1024, 1024), (16, 1024, 1024)]
        if args.mix != 'all':
            mix = [m for m in mix if (m[0] >= 128) == (args.mix == 'long')]
        units = sum(((cout + 127) // 128) * (cin // 64) * 8 * (hw // 8) * (hw // 16) for hw, cin, cout in mix)
        print('mix %s: %d jobs, %d (tile, slab) units = %.1f per workgroup at 256; operands %.0f MB, dW %.0f MB'
              % (args.mix, len(mix), units, units / 256.0, sum(8 * hw * hw * (cin + cout) * 2 for hw, cin, cout in mix) / 1e6,
                 sum(9 * cin * cout * 4 for hw, cin, cout in mix) / 1e6))
        bj = []
        for hw, cin, cout in mix:
            x = torch.randn(8, hw, hw, cin, device=dev).to(torch.bfloat16)
            gy = torch.randn(8, hw, hw, cout, device=dev).to(torch.bfloat16)
            bj.append((x, gy, torch.zeros(cout, 9 * cin, device=dev), torch.zeros(cout, device=dev), None, None))
        flops = sum(2.0 * 8 * hw * hw * cin * cout * 9 for hw, cin, cout in mix)

        def per_layer():
            for x, gy, dw, db, _, _ in bj:
                ops.conv2d_wgrad_raw(x, gy, 3, 3, 1, 1, ops.ACT_NONE, True, db, dw_out=dw)

        last = {}

        def batched():
            last['ws'] = run_batch(bj, dev)
        runs = (('per-layer launches', per_layer), ('one batched launch', batched), ('per-layer launches', per_layer), ('one batched launch', batched))
        if args.only_batched:
            runs = (('one batched launch', batched),)
        for name, fn in runs:
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print('%-20s %7.3f ms  %7.1f TFLOP/s' % (name, ms, flops / ms / 1e9))
        # and the two agree
        for j in bj:
            j[2].zero_(); j[3].zero_()
        per_layer()
        a = [(j[2].clone(), j[3].clone()) for j in bj]
        for j in bj:
            j[2].zero_(); j[3].zero_()
        batched()
        torch.cuda.synchronize()
        for (hw, cin, cout), j, (dw, db) in zip(mix, bj, a):
            e = float((j[2] - dw).abs().max()) / float(dw.abs().max())
            eb = float((j[3] - db).abs().max()) / float(db.abs().max())
            assert e < 1e-4 and eb < 1e-4, (hw, cin, cout, e, eb)
        print('batched == per-layer on the bench mix')
        # the plan workgroup 0 left behind the fragment slots (csrc/conv_wgrad_batch.hip: WbPlan)
        ws = last['ws']
        nwg = (ws.numel() * 4 - 1024) // (3 * 9 * 128 * 64 * 4)
        plan = ws[3 * nwg * 9 * 128 * 64:].view(torch.int32).cpu().numpy()
        w_long, q, u_short, n = [int(v) for v in plan[:4]]
        nb = [int(v) for v in plan[8 + 32:8 + 64][:n]]
        cap = [int(v) for v in plan[8 + 96:8 + 128][:n]]
        wbase = [int(v) for v in plan[8 + 64:8 + 96][:n]]
        print('plan: %d workgroups, quota %d, %d start with a long block, %d stream-K units; blocks per tile %s; first workgroup %s; spare %s'
              % (nwg, q, w_long, u_short, nb, wbase, cap))


if __name__ == '__main__':
    main()
