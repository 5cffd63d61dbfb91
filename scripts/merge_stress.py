#!/usr/bin/env python3
"""bbox-merge stress (BASELINE config C5 shape, scaled by flags): N points in B Gaussian blobs, parent_id = id mod 8.
Prints the wall time of merge_bb (host control flow + one GPU scan per surviving instance)."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))

ap = argparse.ArgumentParser()
ap.add_argument('--points', type=int, default=5_000_000)
ap.add_argument('--boxes', type=int, default=512)
args = ap.parse_args()
from Fusion3DSeg.merge_intersecting_bb import merge_bb  # noqa: E402

rng = np.random.default_rng(3456)
B, n = args.boxes, args.points
centres = rng.uniform([-5, -5, 0], [5, 5, 3], (B, 3))
ids = rng.integers(1, B, n).astype(np.int64)
pts = centres[ids] + rng.normal(size=(n, 3)) * 0.15
info = [{'id': k, 'category_id': 86, 'parent_id': k % 8, 'area': int((ids == k).sum())} for k in range(B)]
t0 = time.perf_counter()
out_info, out_ids = merge_bb(None, info, ids, pts)
dt = time.perf_counter() - t0
print(f'merge_bb: {n} points, {B} instances -> {len(out_info)} in {dt:.2f} s')
