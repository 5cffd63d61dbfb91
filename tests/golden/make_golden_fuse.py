#!/usr/bin/env python3
"""Golden vectors for rows a5 / (f)#2: Fusion.fuse + patch_downsample (Fusion3DSeg/fusion.py:134-324), run from the reference.

Run in the build container: ``python tests/golden/make_golden_fuse.py``.  As in make_golden.py the ``Fusion`` class is
compiled from the reference's file by ``ast`` (its module imports open3d and cv2, which the two methods do not touch);
nothing of it is copied.  An instance is made without the file readers (``object.__new__``): the attributes ``__init__``
would have set (:95-118) are filled from a synthetic 4-frame sequence -- a wall and a floor seen by a camera that moves
sideways -- and ``_save_uv2pt`` is replaced by a collector so that the per-frame lookups, the product the voting stage
consumes, are part of the fixture.  The unseeded ``np.random.shuffle`` of patch_downsample (:172) draws from the global
NumPy generator; it is seeded here, and a drop-in that shuffles at the same places sees the same permutations.
"""
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parent.parent))
sys.path.insert(0, str(OUT))
from make_golden import load_reference  # noqa: E402
from oracle import np_ref as O  # noqa: E402


def synthetic_sequence(h=24, w=32, nframes=4, seed=7):
    """Depth frames of a wall at z = 2.5 m and a floor, camera translating along x: world points [F, h*w, 3], normals,
    colours, validity; intrinsics and (w,x,y,z) poses."""
    rng = np.random.default_rng(seed)
    K = np.array([[30.0, 0.0, w / 2.0], [0.0, 30.0, h / 2.0], [0.0, 0.0, 1.0]])
    q = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (nframes, 1))          # camera axes = world axes (x right, y down, z forward)
    t = np.stack([np.array([0.15 * j, 0.0, 0.0]) for j in range(nframes)])
    pts, nrm, clr, val = [], [], [], []
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    for j in range(nframes):
        dirx, diry = (uu - K[0, 2]) / K[0, 0], (vv - K[1, 2]) / K[1, 1]
        z_wall = np.full((h, w), 2.5)
        with np.errstate(divide='ignore'):
            z_floor = np.where(diry > 1e-9, 0.6 / diry, np.inf)      # floor plane y = 0.6 below the camera
        z = np.minimum(z_wall, z_floor) + rng.normal(0, 0.002, (h, w))
        floor = z_floor < z_wall
        cam = np.stack([dirx * z, diry * z, z], -1).reshape(-1, 3)
        pts.append(O.rotate(q[j], cam) + t[j])
        n = np.where(floor.reshape(-1, 1), np.array([0.0, -1.0, 0.0]), np.array([0.0, 0.0, -1.0]))
        nrm.append(O.rotate(q[j], n))
        clr.append(rng.uniform(0, 1, (h * w, 3)))
        v = np.ones(h * w, bool)
        v[rng.integers(0, h * w, 25)] = False                         # holes in the depth frame
        val.append(v)
    return K, q, t, np.stack(pts), np.stack(nrm), np.stack(clr), np.stack(val)


def main():
    _, _, _, _, Fusion, _ = load_reference()
    h, w, F = 24, 32, 4
    K, q, t, pts, nrm, clr, val = synthetic_sequence(h, w, F)
    out = {'K': K, 'wxyz': q, 't': t, 'points': pts, 'normals': nrm, 'colors': clr, 'valid': val, 'hw': np.array([h, w])}
    cases = [dict(radius=0.05, angle=10, stride=None, max_depth=10, skip=1, seed=11),
             dict(radius=0.08, angle=25, stride=6, max_depth=3.0, skip=1, seed=12),
             dict(radius=0.05, angle=10, stride=4, max_depth=10, skip=2, seed=13)]
    for ci, c in enumerate(cases):
        fu = object.__new__(Fusion)
        fu.K, fu.w, fu.h, fu.xyzws, fu.translations = K, w, h, q, t
        fu.frames = [(f'{100 + j}', pts[j].copy(), nrm[j].copy(), clr[j].copy(), val[j].copy()) for j in range(F)]
        fu.nframes, fu.npts = F, h * w
        fu.ds_radius, fu.ds_angle = None, None
        fu.eyes, fu.lookats, fu.frustum_spoke_origins, fu.frutsum_face_normals = Fusion._get_frustum_data(K, w, h, q, t, np.arange(F))
        fu.pcdimg = np.arange(h * w).reshape(h, w)
        fu.pt2u, fu.pt2v = (np.arange(h * w) % w).astype(np.int32), (np.arange(h * w) // w).astype(np.int32)
        fu.save_lookups = True
        store = {}
        fu._save_uv2pt = lambda uv2pt, name, store=store: store.__setitem__(name, np.array(uv2pt, copy=True))
        np.random.seed(c['seed'])
        ds_pts, ds_norms, ds_clrs, nmerges, occ = fu.fuse(c['radius'], c['angle'], c['stride'], c['max_depth'], c['skip'])
        out[f'c{ci}_params'] = np.array([c['radius'], c['angle'], -1 if c['stride'] is None else c['stride'], c['max_depth'], c['skip'], c['seed']], np.float64)
        out[f'c{ci}_ds_pts'], out[f'c{ci}_ds_norms'], out[f'c{ci}_ds_clrs'] = ds_pts, ds_norms, ds_clrs
        out[f'c{ci}_nmerges'], out[f'c{ci}_occurences'] = np.asarray(nmerges), np.asarray(occ)
        names = sorted(store)
        out[f'c{ci}_uv2pt_names'] = np.array([int(nm) for nm in names])
        out[f'c{ci}_uv2pt'] = np.stack([store[nm] for nm in names])
        print(f'case {ci}: {len(ds_pts)} fused points from {F * h * w}, lookups for frames {names}, '
              f'nmerges sum {int(np.sum(nmerges))}, occurences max {int(np.max(occ))}')
    out['ncases'] = np.array(len(cases))
    np.savez_compressed(OUT / 'fuse.npz', **out)
    print('fuse.npz', (OUT / 'fuse.npz').stat().st_size, 'bytes')


if __name__ == '__main__':
    main()
