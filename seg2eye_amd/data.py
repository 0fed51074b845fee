"""Data sources for train.py / test.py.

The reference trains on the OpenEDS H5 file (data/openeds_dataset.py, data/prepare_openeds.py) -- SURVEY 8 row f4, not
built.  What is built is the data CONTRACT (openeds_dataset.py:103-118: keys `label` (N,1,H,W) int, `style_image`
(N,ns,1,H,W) float in [-1,1], `target` (N,1,H,W) float in [-1,1], `filename`, `user`) served by a seeded synthetic
source (nested-ellipse eye-region labels + smooth images, seg2eye_amd/synthetic.py), one fixed set of samples per
epoch, sharded over data-parallel ranks."""
import torch

from . import synthetic as syn
from .options import image_hw


class SyntheticEyes:
    """len(self) batches per epoch; batch i of epoch e is a pure function of (seed, rank, i)."""

    def __init__(self, opt, rank=0, world=1, seed=1234):
        if opt.dataset_mode != 'synthetic':
            raise NotImplementedError("dataset_mode '%s': 'synthetic' and 'openeds' are built; feed Pix2PixTrainer your own "
                                      "batches with the keys documented in seg2eye_amd/data.py" % opt.dataset_mode)
        self.opt, self.rank, self.world, self.seed = opt, rank, world, seed
        self.h, self.w = image_hw(opt)
        self.n_batches = max(1, int(getattr(opt, 'synthetic_size', 64)) // (opt.batchSize * world))
        self.N = self.n_batches * opt.batchSize

    def __len__(self):
        return self.n_batches

    def batch(self, i):
        b = syn.make_batch(self.opt.batchSize, self.h, self.w, self.opt.input_ns, seed=self.seed + 7919 * (i * self.world + self.rank))
        out = {'label': torch.from_numpy(b['label']), 'style_image': torch.from_numpy(b['style_image']),
               'target': torch.from_numpy(b['target']), 'filename': b['filename'],
               'user': ['synthetic'] * self.opt.batchSize}
        if not getattr(self.opt, 'isTrain', True):
            # the Tester scores against the ORIGINAL 640 x 400 image as 0..255 (openeds_dataset.py:103-118 `target_original`)
            orig = syn.smooth_images('target_original', (self.opt.batchSize, 1, 640, 400), self.seed + 7919 * (i * self.world + self.rank))
            out['target_original'] = torch.from_numpy(((orig + 1.0) * 255.0 / 2.0).astype('int32').astype('uint8'))
        return out

    def __iter__(self):
        for i in range(self.n_batches):
            yield self.batch(i)


def create_dataloader(opt, rank=0, world=1, store=None, style_refs=None):
    """data/__init__.py:43-59.  `--dataset_mode synthetic`: the seeded generator above; `--dataset_mode openeds`: the
    OpenEDS H5 dataset (seg2eye_amd/openeds_dataset.py; `store` = an in-memory H5-like mapping instead of opt.dataroot)."""
    if opt.dataset_mode == 'openeds':
        from .openeds_dataset import create_dataloader as _create
        return _create(opt, store=store, style_refs=style_refs, rank=rank, world=world)
    return SyntheticEyes(opt, rank, world)
