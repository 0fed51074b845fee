#!/usr/bin/env python3
"""Inference / validation entry point, flag-compatible with the reference's test.py (test.py:13-28): loads
`<checkpoints_dir>/<name>/<which_epoch>_net_{G,E}.pth` and hands the model to the Tester --
  * `--dataset_key validation|train` without `--produce_npy`: walks the dataset, scores every generated image with the
    OpenEDS metric (resize to 400 x 640, 0..255, sqrt(sum d^2)/(H W); all on the device) and prints
    `mse/<key>/full/relative`;
  * otherwise: writes one uint8 `.npy` of shape (1, 640, 400) per sample under
    `<checkpoints_dir>/<name>/<results_dir>/<dataset_key>/` plus `pred_npy_list.txt` (util/tester.py:193-219).
Data: `--dataset_mode synthetic` (the OpenEDS H5 pipeline is SURVEY 8 f4)."""
import sys

from seg2eye_amd.options import parse
from seg2eye_amd.pix2pix_model import Pix2PixModel
from seg2eye_amd.tester import Tester


def main(argv=None):
    opt = parse(argv, is_train=False)
    tester = Tester(opt, dataset_key=opt.dataset_key)
    model = Pix2PixModel(opt)
    model.eval()
    limit = opt.how_many if opt.how_many != float('inf') else -1
    if opt.dataset_key in ['validation', 'train'] and not opt.produce_npy:
        all_errors, errors_dict = tester.run(model, mode='full', limit=int(limit), write_error_log=opt.write_error_log)
        return all_errors, errors_dict
    print('Running inference')
    return tester.run_test(model, limit=int(limit))


if __name__ == '__main__':
    main(sys.argv[1:])
