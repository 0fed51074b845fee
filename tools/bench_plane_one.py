"""One conv shape through conv_plane.hip, N launches (for rocprofv3 --pmc runs):  python tools/bench_plane_one.py n h cin cout k s p [dgrad] [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from seg2eye_amd import ops
from seg2eye_amd.ops import conv as oc
n, h, cin, cout, k, s, p = [int(a) for a in sys.argv[1:8]]
dg = len(sys.argv) > 8 and sys.argv[8] == 'dgrad'
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
dt = torch.bfloat16
ho = (h + 2 * p - k) // s + 1
g = torch.Generator().manual_seed(1)
x = torch.randn(n, h, h, cin, generator=g).cuda().to(dt)
gy = torch.randn(n, ho, ho, cout, generator=g).cuda().to(dt)
w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
if not dg:
    pm = oc.plane_mode(dt, n, h, h, cin, ho, ho, cout, k, k, s, p, False)
    wp = oc.pack_weight(w, dt, cin, False, None, plane=pm > 0)
    f = lambda: oc.conv2d_raw(x, wp, None, None, None, (ho, ho, cout), k, k, s, p, False, plane=pm > 0)
else:
    pm = oc.plane_mode(dt, n, ho, ho, cout, h, h, cin, k, k, s, p, True)
    wp = oc.pack_weight(w, dt, cin, True, None, plane=pm > 0)
    f = lambda: oc.conv2d_raw(gy, wp, None, None, None, (h, h, cin), k, k, s, p, True, plane=pm > 0)
for _ in range(reps):
    f()
torch.cuda.synchronize()
print('mode', pm)
