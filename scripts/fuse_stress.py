#!/usr/bin/env python3
"""Fusion.fuse on a synthetic capture (f3d.synth.depth_sequence): wall-clock per frame of the GPU-backed drop-in
(frustum cull + projection, patch-matching ownership and seed-resolution kernels; ordered sums on the host)."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))
from f3d import synth                      # noqa: E402
from Fusion3DSeg.fusion import Fusion      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=12)
ap.add_argument('--height', type=int, default=192)
ap.add_argument('--width', type=int, default=256)
ap.add_argument('--step', type=float, default=0.03, help='camera translation per frame in metres (small = high overlap, as in a real capture)')
args = ap.parse_args()
K, q, t, frames = synth.depth_sequence(args.height, args.width, args.frames, step=args.step)
fu = Fusion.from_frames(K, args.width, args.height, q, t, frames)
np.random.seed(1)
t0 = time.perf_counter()
pts, nrm, clr, nmerges, occ = fu.fuse(0.05, 10, None, 10, 1)
dt = time.perf_counter() - t0
print(f'fuse: {args.frames} frames of {args.height}x{args.width}: {dt:.2f} s = {1e3 * dt / args.frames:.0f} ms per frame; '
      f'{len(pts)} fused points from {args.frames * args.height * args.width} pixels, occurrences up to {int(occ.max())}')
