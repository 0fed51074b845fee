#!/usr/bin/env python3
"""In-kernel phase time stamps of the duo patch kernel (csrc/conv_duo.hip): per tile of three workgroups -- block 0,
block gridDim/2 (its partner on the same CU: tools/probe/dispatch_probe.hip) and block 8 -- the 100-MHz realtime counter at
[top of the tile, after the loop-top wait + barrier, end of the multiply loop, next tile's loads issued, end of the write-out].

The stamps are compiled in only with -DS2E_DUO_STAMPS: build that variant next to the product library and load it through
S2E_LIB_PATH:
    python tools/duo_stamps.py --build          # -> seg2eye_amd/lib/libseg2eye_hip_stamps.so (on the box with hipcc, or here)
    S2E_LIB_PATH=seg2eye_amd/lib/libseg2eye_hip_stamps.so python tools/duo_stamps.py
Prints microseconds: `top` = start of the tile relative to the first stamp, `wait` `loop` `pro` `epi` = the four phases."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_SO = os.path.join(ROOT, 'seg2eye_amd', 'lib', 'libseg2eye_hip_stamps.so')


def build():
    from seg2eye_amd import build as b
    objs = []
    for src in b.SOURCES:
        obj = os.path.join(b.HERE, 'lib', src.replace('.hip', '.o'))
        if src == 'conv_duo.hip':
            obj = os.path.join(b.HERE, 'lib', 'conv_duo_stamps.o')
            subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DS2E_DUO_STAMPS', '-c',
                                   os.path.join(b.CSRC, src), '-o', obj])
        elif not os.path.exists(obj):
            b.build(force=True)
        objs.append(obj)
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', STAMPS_SO] + objs)
    print(STAMPS_SO)


def main():
    if '--build' in sys.argv:
        return build()
    import torch
    dev = torch.device('cuda:0')
    dbg = torch.zeros(3 * 64, dtype=torch.int64, device=dev)
    os.environ['S2E_DUO_DBG_PTR'] = str(dbg.data_ptr())          # (read once, at the first duo launch)
    from seg2eye_amd import ops
    dt = torch.bfloat16

    def timeit(fn, iters=20, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for n, H, cin, cout in ((8, 256, 128, 256), (8, 128, 128, 512), (8, 256, 128, 64)):
        x = torch.randn(n, H, H, cin, device=dev).to(dt)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        wp = ops.pack_weight(w, dt, cin, False)
        f = lambda: ops.conv2d_raw(x, wp, None, None, None, (H, H, cout), 3, 3, 1, 1)
        t = timeit(f)
        fl = 2.0 * n * H * H * cin * cout * 9
        print('c%d->%d @%d: %.1f us %.0f TFLOP/s  (S2E_DUO_MF16=%s)' % (cin, cout, H, t * 1e3, fl / t / 1e9, os.environ.get('S2E_DUO_MF16', '1')), flush=True)
        dbg.zero_()
        f()
        torch.cuda.synchronize()
        rows = dbg.cpu().view(3, -1).tolist()
        if not any(r[0] for r in rows):
            print('  (no stamps: load the -DS2E_DUO_STAMPS build through S2E_LIB_PATH)')
            continue
        t00 = min(r[0] for r in rows if r[0])
        for name, row in zip(('block 0     ', 'block G/2   ', 'block 8     '), rows):
            out = []
            for i in range(12):
                r = row[i * 5:(i + 1) * 5]
                if not r[0]:
                    break
                out.append('[%d: top %.1f wait %.1f loop %.1f pro %.1f epi %.1f]' % (i, (r[0] - t00) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100))
            print(' ', name, ' '.join(out))


if __name__ == '__main__':
    main()
