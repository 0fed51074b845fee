#!/usr/bin/env python3
"""Seconds per G+D step of the CPU oracle at batch 8 (256x256, ngf=ndf=64) for several thread counts on this host
(bench.py's cpu_baseline picks its thread count from this sweep).  CPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

kw = dict(ngf=64, ndf=64, crop_size=256, aspect_ratio=1.0, batchSize=8, compute_dtype='fp32', gpu_ids=[], hip_graphs=False)
print('physical cores', bench._physical_cores(), 'logical', os.cpu_count(), flush=True)
for th in [int(a) for a in sys.argv[1:]] or (128, 64, 32, 16, 8):
    sec, n, w = bench.cpu_step_seconds(kw, 256, 8, th, 1, 2, 120.0)
    print('threads %3d: %.2f s per bs-8 step (%d timed after %d warm-up) = %.3f img/s' % (th, sec, n, w, 8 / sec), flush=True)
