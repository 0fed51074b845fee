import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def manifest_of(z, prefix):
    names = [str(s) for s in z[prefix + '_names']]
    shapes = z[prefix + '_shapes']
    nd = z[prefix + '_ndims']
    return [(n, tuple(int(x) for x in shapes[i][:nd[i]])) for i, n in enumerate(names)]


def filled_state(z, prefix, seed=0):
    """{key: torch tensor} rebuilt from the fixture's manifest by the hash fill."""
    import torch
    from seg2eye_amd.synthetic import fill_state_dict
    sd = fill_state_dict(manifest_of(z, prefix), seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def checksum(t):
    import torch
    t = t.detach().double().flatten().cpu()
    # (fp32 linspace as the first fixtures were written with; past 2^24 elements its end point rounds out of range)
    idx = torch.linspace(0, t.numel() - 1, 16, dtype=torch.float32 if t.numel() <= (1 << 24) else torch.float64).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item()], t[idx].numpy()])


def assert_checksum_close(t, ref, rtol, what='', flip=0.0):
    """`flip` is the size of ONE legitimate discrete difference per element (2*lr*steps for parameters
    stepped by Adam with beta1 = 0: the first update is exactly -lr*sign(g), so a gradient within
    float-summation noise of zero moves its weight by +lr in one implementation and -lr in the other).
    The sums may absorb max(2, 1%) such elements, a sampled element one."""
    got = checksum(t)
    scale = max(ref[1] / max(t.numel(), 1), 1e-12)        # mean |x|
    slack = flip * max(2, 0.01 * t.numel())
    # sum / abs-sum: compare relative to the abs-sum; samples: relative to mean |x|
    assert abs(got[0] - ref[0]) <= rtol * max(ref[1], 1e-12) + slack, (what, got[0], ref[0])
    assert abs(got[1] - ref[1]) <= rtol * max(ref[1], 1e-12) + slack, (what, got[1], ref[1])
    np.testing.assert_allclose(got[2:], ref[2:], rtol=0, atol=rtol * 50 * scale + 1e-9 + flip, err_msg=what)


@pytest.fixture(scope='session')
def golden():
    return load_golden
