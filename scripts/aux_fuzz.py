#!/usr/bin/env python3
"""Randomised parity runs of the paths next to the fused one (GPU box): batched uv2pt vote (with and without an offending frame, negative
lookups, duplicate pixels, patch-structured lookups) against the oracle's frame loop; merge_bb on random blob scenes (prefilter on: blobs
of >= 256 points) against the oracle's literal control flow; segment_votes with random thresholds / filter lists.
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
    want_info, want_ids = O.merge_bb(copy.deepcopy(info), ids.copy(), pts)
    got_info, got_ids = merge_bb(None, copy.deepcopy(info), ids.copy(), pts, box_fn=O.obb_from_points)
    assert np.array_equal(got_ids, want_ids), ('merge ids', nb, per)
    assert [(d['id'], d['area']) for d in got_info] == [(d['id'], d['area']) for d in want_info], ('merge info', nb, per)
    for g, w_ in zip(got_info, want_info):
        assert ('bbox' in g) == ('bbox' in w_) and ('bbox' not in g or np.allclose(g['bbox'], w_['bbox']))


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
        if k % 4 == 0:
            with contextlib.redirect_stdout(io.StringIO()):
                merge_config(rng)
        if k % 50 == 49:
            print(f'{k + 1} configurations ok', flush=True)
    print('all configurations agree with the oracle')
