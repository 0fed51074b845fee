#!/usr/bin/env python3
"""Which Python lines of one eager train step launch torch's own small kernels (copies, fills, element-wise adds)?
    python3 tools/trace_small_ops.py > gpurun_out/small_ops.txt
Everything that matters runs in the HIP library; what is left on torch are a few dozen 3-15 us launches per step, and this
lists them by call site so that they can be folded away one by one."""
import collections
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
    dev = torch.device('cuda', 0)
    opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='bf16', gpu_ids=[0], hip_graphs=False)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Pix2PixTrainer(opt)
    bench.fill_weights(trainer.pix2pix_model)
    data = bench.make_data(8, 256, 1234, dev)

    def step():
        trainer.run_generator_one_step(dict(data))
        trainer.run_discriminator_one_step(dict(data))

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sites = collections.defaultdict(lambda: [0, 0])
    watch = ('fill_', 'zero_', 'add_', 'add', 'copy_', '_to_copy', 'div', 'cat', 'clone', 'zeros', 'mul', 'sum', 'mean')

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__.split('.')[0]
            views = ('view', 'detach', 'as_strided', 'slice', 'select', 'permute', 'transpose', 't', 'expand', 'unsqueeze', 'squeeze', 'alias',
                     '_unsafe_view', 'reshape', 'narrow', 'unbind', 'split', 'chunk', 'empty', 'empty_like', 'empty_strided', 'new_empty',
                     'is_contiguous', 'sym_size', 'sym_stride', 'size', 'stride', 'lift_fresh', 'is_pinned', 'record_stream', '_local_scalar_dense', 'view_as')
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + [out])
            if on_gpu and name not in views:
                numel = 0
                for a in list(args) + [out]:
                    if isinstance(a, torch.Tensor):
                        numel = max(numel, a.numel())
                site = '?'
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if fr.filename.startswith(here) and '/tools/' not in fr.filename:
                        site = '%s:%d %s' % (fr.filename.replace(here + '/', ''), fr.lineno, fr.name)
                        break
                e = sites[(name, site)]
                e[0] += 1
                e[1] += numel
            return out

    with Spy():
        step()
    torch.cuda.synchronize()
    for (name, site), (n, numel) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
        print('x%-3d %12d elems  %-10s %s' % (n, numel, name, site))


if __name__ == '__main__':
    main()
