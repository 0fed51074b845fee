#!/usr/bin/env python3
"""Inference entry point, flag-compatible with the reference's test.py (test.py:13-28): loads
`<checkpoints_dir>/<name>/<which_epoch>_net_{G,E}.pth`, generates one image per sample from its label map and style
images (`Pix2PixModel(data, mode='inference')`) and, with `--produce_npy`, writes them as uint8 `.npy` files of shape
(1, H, W) under `<results_dir>/<name>/` ([-1,1] -> 0..255 with the reference's int truncation, data/postprocessor.py:72).
Data: `--dataset_mode synthetic` (the OpenEDS pipeline and the Tester's resize-to-400x640 metric are SURVEY 8 f3/f4)."""
import os
import sys

import numpy as np
import torch

from seg2eye_amd.data import create_dataloader
from seg2eye_amd.options import parse
from seg2eye_amd.pix2pix_model import Pix2PixModel


def main(argv=None):
    opt = parse(argv, is_train=False)
    model = Pix2PixModel(opt)
    model.eval()
    out_dir = os.path.join(opt.results_dir, opt.name)
    done = 0
    for data_i in create_dataloader(opt):
        if done >= opt.how_many:
            break
        with torch.no_grad():
            fake = model(data_i, mode='inference')
        if opt.produce_npy:
            os.makedirs(out_dir, exist_ok=True)
            img = ((fake.float().cpu() + 1.0) / 2.0 * 255.0).clamp(0, 255).to(torch.int32).numpy().astype(np.uint8)
            for b, fn in enumerate(data_i['filename']):
                np.save(os.path.join(out_dir, os.path.splitext(os.path.basename(fn))[0] + '.npy'), img[b])
        done += fake.shape[0]
    print('generated %d images%s' % (done, (' -> ' + out_dir) if opt.produce_npy else ''))
    return done


if __name__ == '__main__':
    main(sys.argv[1:])
