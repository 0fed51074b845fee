"""Build libseg2eye_hip.so (gfx950) in-tree with hipcc.  No torch headers involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libseg2eye_hip.so')
SOURCES = ['conv_igemm.hip', 'conv_wgrad.hip', 'norm_modulate.hip', 'label_ops.hip', 'loss_adam.hip', 'spectral.hip', 'conv_small.hip', 'conv_c8.hip', 'conv_patch.hip', 'conv_duo.hip', 'conv_plane.hip', 'conv_stream.hip', 'conv_wgrad_patch.hip', 'conv_wgrad_batch.hip', 'conv_wgrad_flat.hip', 'metric.hip', 'style_fc.hip', 'spade_sparse_bwd.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'seg2eye_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, 'lib', src.replace('.hip', '.o'))
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
               os.path.join(CSRC, src), '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
