#!/usr/bin/env python3
"""Same-class connected components (split_into_instances flood fill) on a synthetic 3-D lattice graph:
n = side^3 points, 26-neighbourhood (symmetric), classes in random blobs.  Prints kernel time and edges/s."""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))
import torch  # noqa: E402
import f3d    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--side', type=int, default=160)
args = ap.parse_args()
s = args.side
n = s ** 3
idx = np.arange(n).reshape(s, s, s)
nbr = []
for dx in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dz in (-1, 0, 1):
            sh = np.roll(idx, (dx, dy, dz), axis=(0, 1, 2))         # periodic lattice: every vertex has 27 entries (self included)
            nbr.append(sh.reshape(-1))
nb = np.stack(nbr, axis=1).astype(np.int32).reshape(-1)
offs = np.arange(0, 27 * n + 1, 27, dtype=np.int64)
rng = np.random.default_rng(0)
coarse = rng.integers(0, 6, (s // 8 + 1,) * 3)
cls = np.repeat(np.repeat(np.repeat(coarse, 8, 0), 8, 1), 8, 2)[:s, :s, :s].reshape(-1).astype(np.int64)
ctx = f3d.default_context()
dev = torch.device('cuda', 0)
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    dc, do, dn = (torch.from_numpy(a).to(dev) for a in (cls, offs, nb))
    par = torch.empty(n, dtype=torch.int32, device=dev)
    root = torch.empty(n, dtype=torch.int64, device=dev)
    run = lambda: ctx._check(ctx._lib.f3d_components_same_class_dev(ctx._h, dc.data_ptr(), n, do.data_ptr(), dn.data_ptr(),
                                                                     par.data_ptr(), root.data_ptr(), st.cuda_stream))
    run(); st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(5):
        run()
    b.record(st); b.synchronize()
ms = a.elapsed_time(b) / 5
r = root.cpu().numpy()
ncomp = len(np.unique(r))
assert (r <= np.arange(n)).all() and (cls[r] == cls).all()
print(f'components: {n} points, {27 * n} directed edges, {ncomp} components: {ms:.3f} ms -> {27 * n / ms / 1e6:.1f} G edges/s, '
      f'{(27 * n * 4 + n * 20) / ms / 1e6:.0f} GB/s algorithmic')
