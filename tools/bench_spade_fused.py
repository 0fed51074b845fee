#!/usr/bin/env python3
"""The SPADE+Style forward of every generator layer shape at 256x256 bs=8: [gamma | beta] conv + modulate_fwd as two
launches against s2e_spade_conv_modulate (no-grad: gamma not stored; train: gamma stored).  HIP events on the launch stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seg2eye_amd import ops, _lib as L  # noqa: E402
from bench_kernels import timeit          # noqa: E402


def main():
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
    dev = torch.device('cuda:0')
    N, nh = 8, 128
    tot = [0.0, 0.0, 0.0]
    for h, C, cnt in ((256, 128, 2), (256, 64, 1), (128, 256, 2), (128, 128, 1), (64, 512, 2), (64, 256, 1), (32, 1024, 2), (32, 512, 1),
                      (16, 1024, 4), (8, 1024, 2)):
        x = torch.randn(N, h, h, C, device=dev).to(dt)
        actv = torch.randn(N, h, h, nh, device=dev).relu().to(dt)
        w = torch.randn(2 * C, nh, 3, 3, device=dev) / (nh * 9) ** 0.5
        b = torch.randn(2 * C, device=dev) * 0.1
        wp = ops.pack_weight(w, dt, nh, False)
        style = torch.randn(N, 2 * C, device=dev)
        st = ops.in_stats(x)
        out, gam = torch.empty_like(x), torch.empty_like(x)
        gb = torch.empty(N, h, h, 2 * C, device=dev, dtype=dt)

        def two():
            ops.conv2d_raw(actv, wp, b, None, None, (h, h, 2 * C), 3, 3, 1, 1, out=gb)
            L.check(L.lib().s2e_modulate_fwd(ops._dt(x), 0, x.data_ptr(), gb.data_ptr(), st.data_ptr(), style.data_ptr(), out.data_ptr(),
                                             N, h * h, C, 1, 0, ops._stream()))

        def fused(g):
            L.check(L.lib().s2e_spade_conv_modulate(ops._dt(x), actv.data_ptr(), wp.data_ptr(), b.data_ptr(), x.data_ptr(), st.data_ptr(),
                                                    style.data_ptr(), 0, out.data_ptr(), g.data_ptr() if g is not None else None,
                                                    N, h, h, C, nh, 1, 1, ops._stream()))
        t2 = timeit(two)
        tc = timeit(lambda: ops.conv2d_raw(actv, wp, b, None, None, (h, h, 2 * C), 3, 3, 1, 1, out=gb))
        tn = timeit(lambda: fused(None))
        tt = timeit(lambda: fused(gam))
        sup = ops.spade_fused_supported(x, nh)
        gf = 2.0 * N * h * h * nh * 2 * C * 9 / 1e9
        print('%3dx%-3d C=%-4d x%d  conv %.3f ms (%4.0f TF) +mod = %.3f | fused nograd %.3f (%4.0f TF)  train %.3f (%4.0f TF) %s'
              % (h, h, C, cnt, tc, gf / tc, t2, tn, gf / tn, tt, gf / tt, '' if sup else '[not taken by default]'), flush=True)
        if sup:
            tot[0] += cnt * 2 * t2; tot[1] += cnt * tn; tot[2] += cnt * tt
    print('per step over the taken layers: two-launch %.3f ms, fused %.3f ms (one no-grad + one train forward)' % (tot[0], tot[1] + tot[2]))


if __name__ == '__main__':
    main()
