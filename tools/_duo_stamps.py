import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda:0')
dbg = torch.zeros(3 * 64, dtype=torch.int64, device=dev)
os.environ['S2E_DUO_DBG_PTR'] = str(dbg.data_ptr())
from seg2eye_amd import ops, _lib as L
dt = torch.bfloat16
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n, H, cin, cout in ((8, 256, 128, 256), (8, 128, 128, 512)):
    x = torch.randn(n, H, H, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wp = ops.pack_weight(w, dt, cin, False)
    f = lambda: ops.conv2d_raw(x, wp, None, None, None, (H, H, cout), 3, 3, 1, 1)
    t = timeit(f)
    fl = 2.0 * n * H * H * cin * cout * 9
    print('duo=%s grid=%s stagger=%s c%d->%d @%d: %.1f us %.0f TF' % (os.environ.get('S2E_CONV_DUO'), os.environ.get('S2E_DUO_GRID'), os.environ.get('S2E_DUO_STAGGER'), cin, cout, H, t * 1e3, fl / t / 1e9), flush=True)
    dbg.zero_(); f(); torch.cuda.synchronize()
    d = dbg.cpu().view(3, -1).tolist()
    t00 = min(r[0] for r in d if r[0])
    for name, row in zip(('blk0', 'blkG/2', 'blk8'), d):
        out = []
        for i in range(12):
            r = row[i * 5:(i + 1) * 5]
            if not r[0]: break
            out.append('[%d: top %.1f wait %.1f loop %.1f pro %.1f epi %.1f]' % (i, (r[0] - t00) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, (r[4] - r[3]) / 100))
        print(' ', name, ' '.join(out))
