"""Deterministic synthetic inputs and weight fill (host side, numpy only).

Nothing here depends on torch's RNG streams: every value is a pure function of
(name, flat index, seed) through a 64-bit integer hash, so the golden-vector
generator (which imports the reference), the oracle tests, the GPU parity
tests and bench.py all see bit-identical inputs on any machine
("deterministic fill keyed by state-dict name").
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def hash_uniform(name, shape, seed=0, lo=-1.0, hi=1.0):
    """float32 array in [lo, hi), a pure function of (name, index, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) ^ (int(seed) * 0x9E3779B1 & 0xFFFFFFFF))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (key << np.uint64(32))
        h = _splitmix64(idx)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)     # 24 random bits
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def hash_normal(name, shape, seed=0):
    """Approximately N(0,1) float32 (Irwin-Hall sum of 4 uniforms, rescaled)."""
    acc = np.zeros(shape, np.float64)
    for k in range(4):
        acc += hash_uniform('%s#%d' % (name, k), shape, seed, -1.0, 1.0).astype(np.float64)
    return (acc / np.sqrt(4.0 / 3.0)).astype(np.float32)


def fill_state_entry(name, shape, seed=0):
    """Trained-scale deterministic value for one state-dict entry, chosen by
    the key's suffix (pin weight_orig, weight_u, weight_v and use
    weights large enough that the generator output has std ~0.3-0.5)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf in ('weight_u', 'weight_v'):
        v = hash_normal(name, shape, seed).astype(np.float64)
        return (v / max(np.linalg.norm(v), 1e-12)).astype(np.float32)
    if leaf == 'bias':
        return hash_uniform(name, shape, seed, -0.1, 0.1)
    if leaf in ('weight', 'weight_orig'):
        if len(shape) == 4:                       # conv: (Cout, Cin, kh, kw)
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.0
            if 'mlp_gamma' in name or 'mlp_beta' in name:
                gain = 0.7
            if 'conv_img' in name:
                gain = 2.0
            return (hash_normal(name, shape, seed) * (gain / np.sqrt(fan_in))).astype(np.float32)
        if len(shape) == 2:                       # FC / Linear: (out, in)
            if 'adain' in name:
                return (hash_normal(name, shape, seed) * 0.25).astype(np.float32)
            return (hash_normal(name, shape, seed) / np.sqrt(shape[1])).astype(np.float32)
        if len(shape) == 1:                       # BatchNorm2d(affine=True).weight (norm_D / norm_E spectralbatch): scale near 1
            return hash_uniform(name, shape, seed, 0.7, 1.3)
    if leaf == 'running_mean':                        # BatchNorm SPADE buffers: plausible non-trivial running statistics
        return hash_uniform(name, shape, seed, -0.2, 0.2)
    if leaf == 'running_var':
        return hash_uniform(name, shape, seed, 0.5, 1.5)
    if leaf == 'num_batches_tracked':
        return np.full(shape, 3, dtype=np.int64)
    raise ValueError('no fill rule for state entry %r with shape %r' % (name, shape))


def settle_spectral(sd, iters=30):
    """Run ``iters`` power iterations (float64) on every (weight_orig, weight_u,
    weight_v) triple so u, v sit at the top singular pair, as they do in a
    trained checkpoint.  From random u, v the reference's train-mode forward
    (one more iteration per call) is wildly off-scale."""
    for k in list(sd):
        if not k.endswith('.weight_orig'):
            continue
        base = k[:-len('weight_orig')]
        wm = sd[k].reshape(sd[k].shape[0], -1).astype(np.float64)
        u = sd[base + 'weight_u'].astype(np.float64)
        v = sd[base + 'weight_v'].astype(np.float64)
        for _ in range(iters):
            v = wm.T @ u
            v /= max(np.linalg.norm(v), 1e-12)
            u = wm @ v
            u /= max(np.linalg.norm(u), 1e-12)
        sd[base + 'weight_u'] = u.astype(np.float32)
        sd[base + 'weight_v'] = v.astype(np.float32)
    return sd


def fill_state_dict(manifest, seed=0, settle=True):
    """manifest: iterable of (name, shape) -> {name: float32 ndarray}."""
    sd = {name: fill_state_entry(name, shape, seed) for name, shape in manifest}
    return settle_spectral(sd) if settle else sd


# --------------------------------------------------------------------------- data

def ellipse_labels(n, h, w, seed=1234):
    """(n,1,h,w) uint8 4-class maps: background 0 > sclera 1 > iris 2 > pupil 3
    as nested ellipses with per-sample centre/radii."""
    out = np.zeros((n, 1, h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    r = hash_uniform('ellipse_labels', (n, 6), seed, 0.0, 1.0)
    for i in range(n):
        cy = (0.40 + 0.20 * r[i, 0]) * h
        cx = (0.40 + 0.20 * r[i, 1]) * w
        ry = (0.22 + 0.10 * r[i, 2]) * h
        rx = (0.34 + 0.12 * r[i, 3]) * w
        d_scl = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        ri = (0.45 + 0.15 * r[i, 4]) * min(ry, rx)
        d_iris = ((yy - cy) ** 2 + (xx - cx) ** 2) / (ri * ri)
        rp = (0.30 + 0.25 * r[i, 5]) * ri
        d_pup = ((yy - cy) ** 2 + (xx - cx) ** 2) / (rp * rp)
        m = out[i, 0]
        m[d_scl < 1.0] = 1
        m[(d_iris < 1.0) & (d_scl < 1.0)] = 2
        m[(d_pup < 1.0) & (d_scl < 1.0)] = 3
    return out


def _box5(a):
    """5x5 box filter with edge replication over the last two dims."""
    p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(2, 2), (2, 2)], mode='edge').astype(np.float64)
    c = np.cumsum(np.cumsum(p, axis=-2), axis=-1)
    c = np.pad(c, [(0, 0)] * (a.ndim - 2) + [(1, 0), (1, 0)])
    h, w = a.shape[-2:]
    s = c[..., 5:5 + h, 5:5 + w] - c[..., 0:h, 5:5 + w] - c[..., 5:5 + h, 0:w] + c[..., 0:h, 0:w]
    return (s / 25.0).astype(np.float32)


def smooth_images(name, shape, seed=1234):
    """uniform[-1,1] noise smoothed by a 5x5 box, rescaled to use [-1,1]."""
    a = _box5(hash_uniform(name, shape, seed, -1.0, 1.0))
    return np.clip(a * 2.5, -1.0, 1.0).astype(np.float32)


def make_batch(n, h, w, input_ns=4, seed=1234):
    """The reference's data dict (data/openeds_dataset.py:103-118) with 4-D
    labels, as numpy arrays."""
    return {
        'label': ellipse_labels(n, h, w, seed),                               # (n,1,h,w) uint8
        'style_image': smooth_images('style_image', (n, input_ns, 1, h, w), seed),
        'target': smooth_images('target', (n, 1, h, w), seed),
        'filename': ['synthetic_%d_%d' % (seed, i) for i in range(n)],
    }
