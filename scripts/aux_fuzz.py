#!/usr/bin/env python3
"""Randomised parity runs of the paths next to the fused one (GPU box): batched uv2pt vote (with and without an offending frame, negative
lookups, duplicate pixels, patch-structured lookups) against the oracle's frame loop; merge_bb on random blob scenes (prefilter on: blobs
of >= 256 points; the oracle's fit injected or the product's own GPU fit) against the oracle's literal control flow; segment_votes with random
thresholds / filter lists; f3d_obb_fit on random point sets (hull vertices against scipy's Qhull, boxes against the oracle's recipe).
usage: scripts/aux_fuzz.py [--configs N] [--seed S]"""
import argparse
import copy
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))
sys.path.insert(0, str(ROOT))
import f3d                                  # noqa: E402
from oracle import np_ref as O              # noqa: E402


def vote_config(ctx, rng):
    npts = int(rng.choice([3, 50, 500, 20_000]))
    h, w = int(rng.integers(1, 80)), int(rng.integers(1, 90))
    F = int(rng.integers(1, 12))
    ncols = int(rng.choice([1, 4, 134]))
    kind = rng.integers(3)
    if kind == 0:
        luts = rng.integers(-1, npts, (F, h * w))
    elif kind == 1:                                              # patches: many pixels of a frame share a point
        luts = np.repeat(np.repeat(rng.integers(-1, npts, (F, (h + 4) // 5, (w + 4) // 5)), 5, axis=1), 5, axis=2)[:, :h, :w].reshape(F, -1)
    else:                                                        # NumPy negative indices
        luts = rng.integers(-npts, npts, (F, h * w))
        luts[luts == -1] = 0
    luts = luts.astype(np.int32)
    masks = rng.integers(0, ncols, (F, h * w)).astype(np.uint8)
    fbad = None
    if rng.random() < 0.4:
        fbad = int(rng.integers(0, F)); px = int(rng.integers(0, h * w))
        if rng.random() < 0.5 and ncols < 255:
            masks[fbad, px] = ncols; luts[fbad, px] = 0
        else:
            luts[fbad, px] = npts if rng.random() < 0.5 else -npts - 1
    want = rng.integers(0, 3, (npts, ncols)).astype(np.float64)
    got = want.copy()
    for f in range(F if fbad is None else fbad):
        O.vote_frame(want, luts[f], masks[f])
    try:
        ctx.vote_uv2pt_batch(got, luts, masks, h, w)
        assert fbad is None, 'no IndexError'
    except IndexError:
        assert fbad is not None, 'spurious IndexError'
    assert np.array_equal(got, want), ('vote', npts, h, w, F, ncols, kind, fbad)
    thr = float(rng.choice([0.0, 0.25, 0.5, 0.75, 1.0]))
    flt = None if rng.random() < 0.5 or ncols < 2 else list(map(int, rng.integers(0, ncols, int(rng.integers(1, 6)))))
    assert np.array_equal(ctx.segment_votes(got, ncols - 1, thr, flt), O.segment(got, ncols - 1, thr, flt)), ('segment', ncols, thr, flt)


def merge_config(rng):
    from Fusion3DSeg.merge_intersecting_bb import merge_bb
    nb = int(rng.integers(6, 40))
    per = int(rng.choice([30, 120, 400]))                        # 400: the GPU inner-hull prefilter takes part
    centres = rng.uniform(-2, 2, (nb, 3)) * float(rng.choice([0.4, 1.0, 2.5]))
    ids = np.repeat(np.arange(nb), per)
    pts = centres[ids] + rng.normal(size=(len(ids), 3)) * 0.2
    order = rng.permutation(len(ids)); ids, pts = ids[order].astype(np.int64), pts[order]
    for s_ in rng.choice(np.arange(3, nb), 2, replace=False):    # instances with < 4 points: early return (:83-84)
        idx = np.nonzero(ids == s_)[0]
        ids[idx[2:]] = int(rng.integers(1, 3))
    info = [{'id': k, 'category_id': 86, 'parent_id': int(rng.integers(0, 3)), 'area': int((ids == k).sum())} for k in range(nb)]
    # alternately: the control flow with the oracle's fit injected (on all members / on the hull candidates), and the product's own GPU fit.
    # The reference's decision "the two boxes share a cloud point" hangs on points that lie ON a box face -- the very vertices that define a
    # box's extents are inside or outside it by the last bit of whoever computed the box (measured: ~2 such points per box between the GPU fit
    # and the LAPACK recipe, all within 3 ulp of a face; Open3D's own rounding would be a third opinion).  So the product's own fit is compared
    # with the oracle's control flow GIVEN THE SAME BOXES (the oracle calls the GPU fit per instance, on all members), bit for bit; how close
    # those boxes are to the oracle's recipe is obb_config's business.
    mode = int(rng.integers(3))
    ctx = f3d.default_context()

    def gpu_fit(p):
        boxes, status = ctx.obb_fit([p])
        if status[0] != f3d.OBB_OK:
            return O.obb_from_points(p)
        return boxes[0, 0:3].copy(), boxes[0, 3:12].reshape(3, 3).copy(), boxes[0, 12:15].copy()
    want_info, want_ids = O.merge_bb(copy.deepcopy(info), ids.copy(), pts, **({'box_fn': gpu_fit} if mode == 2 else {}))
    kw = [dict(box_fn=O.obb_from_points), dict(box_fn=O.obb_from_points, prefilter=True), dict()][mode]
    got_info, got_ids = merge_bb(None, copy.deepcopy(info), ids.copy(), pts, **kw)
    assert np.array_equal(got_ids, want_ids), ('merge ids', nb, per, mode)
    assert [(d['id'], d['area']) for d in got_info] == [(d['id'], d['area']) for d in want_info], ('merge info', nb, per, mode)
    for g, w_ in zip(got_info, want_info):
        assert ('bbox' in g) == ('bbox' in w_), mode
        if 'bbox' in g:
            assert np.array_equal(np.array(g['bbox']), np.array(w_['bbox'])), ('bbox', mode)      # the same fit on both sides: the same bits


def obb_config(ctx, rng):
    """f3d_obb_fit on random point sets: hull vertex set == scipy's Qhull for every set the kernel certifies, box == the oracle's recipe as a
    corner set; sets with exact degeneracies (duplicates, coplanar quadruples on a lattice) must come back deferred, never wrong."""
    from scipy.spatial import ConvexHull, QhullError
    sets = []
    for _ in range(int(rng.integers(1, 12))):
        m = int(rng.choice([4, 5, 7, 30, 200, 900]))
        kind = int(rng.integers(4))
        if kind == 0:
            p = rng.normal(size=(m, 3)) * rng.uniform(0.01, 3.0, 3)
        elif kind == 1:
            p = rng.uniform(-1, 1, (m, 3)) * rng.uniform(0.1, 50.0)
        elif kind == 2:                                          # a coarse lattice: duplicates and coplanar quadruples are certain
            p = rng.integers(0, 6, (m, 3)).astype(np.float64) * 0.25
        else:                                                    # nearly flat
            p = rng.normal(size=(m, 3)) * [1.0, 1.0, 1e-7]
        sets.append(p @ np.linalg.qr(rng.normal(size=(3, 3)))[0].T + rng.uniform(-10, 10, 3) if kind != 2 else p)
    boxes, status, verts = ctx.obb_fit(sets, want_vertices=True)
    for k, p in enumerate(sets):
        if status[k] != f3d.OBB_OK:
            assert (boxes[k] == 0).all()
            continue
        try:
            hv = np.sort(ConvexHull(p).vertices)
        except QhullError:
            raise AssertionError(('certified a set Qhull calls degenerate', k))
        assert np.array_equal(np.flatnonzero(verts[k]), hv), ('hull vertices', k, len(p))
        c, R, e = O.obb_from_points(p)
        gc, gR, ge = boxes[k, 0:3], boxes[k, 3:12].reshape(3, 3), boxes[k, 12:15]
        from Fusion3DSeg.merge_intersecting_bb import obb_corners
        a, b = obb_corners(gc, gR, ge), obb_corners(c, R, e)
        ev = np.linalg.eigvalsh(np.cov((p[verts[k]] - p[verts[k]].mean(0)).T, bias=True))
        gap = np.diff(np.sort(ev)).min() / max(ev.max(), 1e-300)
        if gap > 1e-3:                                           # (close eigenvalues: the axes are ill-conditioned in any solver)
            d = np.abs(a[:, None, :] - b[None, :, :]).max(-1)
            tol = 1e-9 * (np.abs(p).max() + 1) / gap
            assert (d.min(1) < tol).all() and (d.min(0) < tol).all(), ('box', k, len(p), gap)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', type=int, default=200)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = f3d.Context(0)
    import contextlib, io
    for k in range(a.configs):
        vote_config(ctx, rng)
        if k % 2 == 0:
            obb_config(ctx, rng)
        if k % 4 == 0:
            with contextlib.redirect_stdout(io.StringIO()):
                merge_config(rng)
        if k % 50 == 49:
            print(f'{k + 1} configurations ok', flush=True)
    print('all configurations agree with the oracle')
