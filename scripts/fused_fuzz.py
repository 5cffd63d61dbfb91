#!/usr/bin/env python3
"""Randomised parity run of the fused path (GPU box): scenes of random size, view count (incl. > 64 and > 255), mask size (odd widths,
masks of another size than the frustum was built for), intrinsics, label alphabets (every k_fuse instance), thresholds, filter lists,
dtypes, sort flags and chunk splits -- labels (and vote rows) against the oracle on a random subset, the chunked call against the
one-shot call on every point.  usage: scripts/fused_fuzz.py [--configs N] [--seed S]"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))
sys.path.insert(0, str(ROOT))
import f3d                                  # noqa: E402
from f3d import synth                       # noqa: E402
from oracle import np_ref as O              # noqa: E402


def one_config(ctx, rng, k, verbose=True):
    import torch
    dev = torch.device('cuda', 0)
    n = int(rng.choice([1, 63, 129, 700, 5_000, 40_000, 150_000, 400_000]))
    V = int(rng.choice([1, 2, 7, 16, 33, 64, 65, 100, 130, 240, 247, 255, 256, 300]))   # 240..255: the 8-bit "no sample" byte boundary
    if V > 130:
        n = min(n, 40_000)
    W = int(rng.choice([64, 100, 333, 512, 640, 1001])); H = int(rng.choice([48, 100, 240, 512, 777]))
    f = float(rng.uniform(0.5, 1.5) * max(W, H))
    K = np.array([[f, 0, W / 2 + rng.uniform(-5, 5)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-5, 5)], [0, 0, 1]])
    q, t = synth.ring_views(V, seed=int(rng.integers(1 << 30)))
    nlabels = int(rng.choice([1, 3, 9, 11, 12, 30, 47, 60, 98, 99, 134]))
    alphabet = rng.choice(134, nlabels, replace=False).astype(np.uint8)
    blk = int(rng.choice([1, 8, 37, 64]))
    bh, bw = (H + blk - 1) // blk, (W + blk - 1) // blk
    masks = alphabet[rng.integers(0, nlabels, (V, bh, bw))]
    masks = np.ascontiguousarray(np.repeat(np.repeat(masks, blk, axis=1), blk, axis=2)[:, :H, :W])
    max_depth = float(rng.choice([3.0, 10.0]))
    fw, fh = (W, H) if rng.random() < 0.8 else (W + 16, H - 8)       # frustum built for another image size: range test path
    views = f3d.views_build(K, fw, fh, q, t, max_depth)
    pts = synth.cloud(n, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.3:
        pts[rng.integers(0, n, max(1, n // 50))] *= rng.choice([1e31, np.nan, 1e-30])
    thr = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    flt = [None, [86, 114, 115], [int(alphabet[0])], list(map(int, rng.choice(134, 12)))][int(rng.integers(4))]
    f32 = bool(rng.random() < 0.3)
    flags = int(rng.choice([0, f3d.FUSE_SORT]))
    x = torch.from_numpy(pts.astype(np.float32) if f32 else pts).to(dev)
    vd, md = torch.from_numpy(views).to(dev), torch.from_numpy(masks).to(dev)
    cls = torch.full((n,), -7, dtype=torch.int64, device=dev)
    want_votes = V <= 300 and rng.random() < 0.5
    votes = torch.zeros((n, 134), dtype=torch.uint16, device=dev) if want_votes else None
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))        # the fills / copies above run on the current stream
    dt = f3d.F32 if f32 else f3d.F64
    ctx.project_vote_argmax_dev(x.data_ptr(), dt, n, vd.data_ptr(), V, md.data_ptr(), H, W, 133, thr, flt, cls.data_ptr(),
                                None if votes is None else votes.data_ptr(), s.cuda_stream, flags=flags)
    ctx.take_device_error(s.cuda_stream)
    s.synchronize()
    one = cls.cpu().numpy()
    sub = rng.choice(n, min(n, 1500), replace=False)
    p_sub = (pts[sub].astype(np.float32).astype(np.float64) if f32 else pts[sub])
    wviews_K = K
    # the oracle takes the frustum's image size separately from the mask's
    want, wv = O.project_vote_argmax(p_sub, wviews_K, q, t, masks, max_depth, 133, thr, flt, return_votes=True) if (fw, fh) == (W, H) else (None, None)
    if want is None:
        votes_ref = O.forward_votes(p_sub, K, q, t, masks, max_depth, w=fw, h=fh, ncols=134)
        want, wv = O.segment(votes_ref, 133, thr, flt), votes_ref
    assert np.array_equal(one[sub], want), ('labels', k)
    if votes is not None:
        rows = votes.view(torch.int16)[torch.from_numpy(sub).to(dev)].cpu().numpy().view(np.uint16)
        assert np.array_equal(rows.astype(np.float64), wv), ('votes', k)
    chunked = None
    if V <= 255:
        cuts = sorted(set([0, V] + list(map(int, rng.integers(1, V + 1, int(rng.integers(0, 4)))))))
        present = torch.empty(256, dtype=torch.uint8, device=dev)
        ctx.mask_presence_dev(md.data_ptr(), V, H, W, present.data_ptr(), s.cuda_stream)
        ctx.fuse_chunked_begin_dev(present.data_ptr() if rng.random() < 0.7 else None, n, V, H, W, 133, flt, s.cuda_stream)
        cls.fill_(-7)
        s.wait_stream(torch.cuda.current_stream(dev))
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.fuse_chunk_dev(x.data_ptr(), dt, n, vd.data_ptr(), V, a, b, md.data_ptr(), H, W, 133, thr, flt, cls.data_ptr(), s.cuda_stream, flags=flags)
        ctx.take_device_error(s.cuda_stream)
        s.synchronize()
        chunked = cls.cpu().numpy()
        assert np.array_equal(chunked, one), ('chunked', k, cuts)
    if verbose:
        d = ctx.fuse_deferred(s.cuda_stream)
        print(f'{k:3d} ok  n={n} V={V} {H}x{W} frustum {fh}x{fw} labels={nlabels} blk={blk} thr={thr} flt={flt if flt is None else len(flt)} '
              f'f32={f32} flags={flags} votes={want_votes} chunks={None if chunked is None else len(cuts) - 1} real={float((want != 133).mean()):.2f} deferred={d}', flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', type=int, default=60)
    ap.add_argument('--seed', type=int, default=2024)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = f3d.Context(0)
    for k in range(a.configs):
        one_config(ctx, rng, k)
    print('all configurations agree with the oracle')
