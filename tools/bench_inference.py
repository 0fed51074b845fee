#!/usr/bin/env python3
"""G-only inference rate (the second half of BASELINE.json's north_star): Pix2PixModel(mode='inference') = netE on the
style images + netG in eval mode (no power iteration), then the validation post-processing on the device (resize to
400x640, 0..255 truncation), batch 8, 256x256, bf16, inputs resident in HBM.
    python3 tools/bench_inference.py [--batch 8] [--steps 50]"""
import argparse
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--dtype', default='bf16')
    args = ap.parse_args()
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_model import Pix2PixModel
    from seg2eye_amd.postprocessor import ImageProcessor
    dev = torch.device('cuda', 0)
    opt = default_opt(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=args.batch, compute_dtype=args.dtype, gpu_ids=[0])
    # (a train-mode opt: a test-mode model would look for a checkpoint; the weights are the bench's seeded fill either way)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Pix2PixModel(opt)
    bench.fill_weights(model)
    model.eval()
    data = bench.make_data(args.batch, 256, 1234, dev)

    def once():
        with torch.no_grad():
            fake = model.forward(dict(data), mode='inference')
            return ImageProcessor.to_255resized_imagebatch(fake)
    for _ in range(5):
        out = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print('inference: %.1f img/s (%.2f ms per batch of %d, %s; output %s %s)'
          % (args.batch / dt, dt * 1e3, args.batch, args.dtype, tuple(out.shape), out.dtype))


if __name__ == '__main__':
    main()
