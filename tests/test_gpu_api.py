"""C-ABI contract on a real device: allocation-free _dev calls after f3d_ctx_reserve, strict contexts, per-operation
device error bits, the largest box counts of f3d_points_in_obb, and the point-sharded step with the HIP path under a
two-rank process group (both ranks on device 0, gloo)."""
import os

import numpy as np
import pytest

import f3d
from f3d import synth
from oracle import np_ref as O

pytestmark = pytest.mark.gpu


def _free_port():
    """A TCP port nobody listens on right now (a fixed rendezvous port can still be held by an earlier run's socket)."""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def _scene(n=50_000):
    sc = synth.scene('C1', n=n)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    return sc, views


def test_reserved_context_does_not_allocate_and_strict_context_refuses_to():
    import torch
    dev = torch.device('cuda', 0)
    sc, views = _scene()
    n, (V, H, W) = len(sc['points']), sc['masks'].shape
    x, vd, md = (torch.from_numpy(a).to(dev) for a in (sc['points'], views, sc['masks']))
    cls = torch.empty(n, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream(dev)

    def call(ctx, npts=n):
        ctx.project_vote_argmax_dev(x.data_ptr(), f3d.F64, npts, vd.data_ptr(), V, md.data_ptr(), H, W, 133, 0.5, None,
                                    cls.data_ptr(), None, s.cuda_stream, flags=f3d.FUSE_SORT)
        s.synchronize()

    ctx = f3d.Context(0)
    ctx.reserve(n, V, H, W)
    before = ctx.alloc_count
    assert before > 0
    call(ctx); call(ctx); call(ctx, n // 2)
    assert ctx.alloc_count == before                       # no hipMalloc inside the calls
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'])
    assert np.array_equal(cls.cpu().numpy()[:n // 2], want[:n // 2])
    ctx.close()

    strict = f3d.Context(0)
    strict.set_strict(True)
    with pytest.raises(MemoryError):
        call(strict)                                       # nothing reserved: F3D_ERR_NOMEM instead of a hidden hipMalloc
    strict.reserve(n, V, H, W)
    call(strict)
    assert np.array_equal(cls.cpu().numpy(), want)
    taller = torch.zeros((V, 2 * H, W), dtype=torch.uint8, device=dev)
    with pytest.raises(MemoryError):                       # a larger problem than reserved
        strict.project_vote_argmax_dev(x.data_ptr(), f3d.F64, n, vd.data_ptr(), V, taller.data_ptr(), 2 * H, W, 133, 0.5, None,
                                       cls.data_ptr(), None, s.cuda_stream, flags=0)
    strict.close()

    lazy = f3d.Context(0)                                  # the default: scratch grows on first use, then stays
    call(lazy)
    grown = lazy.alloc_count
    call(lazy)
    assert lazy.alloc_count == grown
    lazy.close()


def test_device_error_bits_belong_to_their_operation():
    """A stale IndexError flag of the fused path must neither skip nor be blamed on a later uv2pt vote (and vice versa)."""
    import torch
    dev = torch.device('cuda', 0)
    ctx = f3d.Context(0)
    sc, views = _scene(5000)
    bad = sc['masks'].copy(); bad[:] = 200
    n, (V, H, W) = len(sc['points']), bad.shape
    x, vd, md = (torch.from_numpy(a).to(dev) for a in (sc['points'], views, bad))
    cls = torch.empty(n, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream(dev)
    ctx.project_vote_argmax_dev(x.data_ptr(), f3d.F64, n, vd.data_ptr(), V, md.data_ptr(), H, W, 133, 0.5, None, cls.data_ptr(), None, s.cuda_stream)
    s.synchronize()                                        # the fused error is now pending in the context, nobody has taken it
    votes = np.zeros((6, 3))
    ctx.vote_uv2pt(votes, np.array([0, 1, 5, 5], np.int32), np.array([0, 2, 1, 1], np.uint8))
    assert votes.sum() == 3 and votes[5, 1] == 1           # the vote ran (it used to be skipped silently)
    with pytest.raises(IndexError, match='vote_uv2pt'):
        ctx.vote_uv2pt(votes, np.array([0, 9], np.int32), np.array([0, 0], np.uint8))
    assert votes.sum() == 3
    with pytest.raises(IndexError, match='project_vote_argmax'):
        ctx.take_device_error(s.cuda_stream)
    ctx.take_device_error(s.cuda_stream)                   # consumed
    offs = np.array([0, 1, 2], np.int64)
    with pytest.raises(IndexError, match='components_same_class'):
        ctx.components_same_class(np.array([1, 1]), offs, np.array([1, 7], np.int32))
    ctx.vote_uv2pt(votes, np.array([2], np.int32), np.array([2], np.uint8))
    assert votes[2, 2] == 1
    ctx.close()


@pytest.mark.parametrize('B', [37, 2047, 4096])
def test_points_in_obb_large_and_ragged_box_counts(B):
    """Box counts that are not a multiple of 32 and the maximum of one call (4096 boxes = 128 KiB of LDS bitset per block,
    beyond the default dynamic-LDS limit: the launch has to raise it and report a failure to do so)."""
    ctx = f3d.default_context()
    rng = np.random.default_rng(B)
    pts = rng.uniform([-5, -5, 0], [5, 5, 3], (20_000, 3))
    boxes = np.zeros((B, 15))
    boxes[:, 0:3] = rng.uniform([-5, -5, 0], [5, 5, 3], (B, 3))
    for k in range(B):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        boxes[k, 3:12] = q.reshape(-1)
    boxes[:, 12:15] = rng.uniform(0.2, 0.9, (B, 3))
    inside, cooc = ctx.points_in_obb(pts, boxes)
    sel = rng.choice(B, 24, replace=False)
    for k in sel:
        assert np.array_equal(inside[:, k], O.points_in_obb(pts, boxes[k, 0:3], boxes[k, 3:12].reshape(3, 3), boxes[k, 12:15])), k
    m = inside.astype(np.float32)
    assert np.array_equal(cooc, (m.T @ m) > 0)
    assert inside.any(0).mean() > 0.5
    # boxes whose R is NOT orthonormal (sheared, scaled, one singular, one NaN), huge and tiny extents: the float32 bounds of the
    # pre-test must never hide a point the in-box test accepts
    odd = boxes[:16].copy()
    for k in range(16):
        odd[k, 3:12] = (boxes[k, 3:12].reshape(3, 3) @ (np.eye(3) + rng.normal(size=(3, 3)) * 0.4) * rng.uniform(0.3, 3.0)).reshape(-1)
    odd[3, 3:12] = np.array([[1, 2, 3], [2, 4, 6], [0, 0, 1.0]]).reshape(-1)          # singular
    odd[5, 12:15] = [1e6, 1e-9, 2.0]
    odd[7, 3] = np.nan
    with np.errstate(all='ignore'):
        inside2, _ = ctx.points_in_obb(pts, odd)
        for k in range(16):
            assert np.array_equal(inside2[:, k], O.points_in_obb(pts, odd[k, 0:3], odd[k, 3:12].reshape(3, 3), odd[k, 12:15])), k
    assert inside2[:, :3].any()


@pytest.mark.parametrize('B', [8, 33, 64])
def test_points_in_obb_cell_table_path(B):
    """8 .. 64 boxes per call: a point visits only the boxes registered in its cell of a 2048-cell grid over the boxes' bounds
    (k_obb_cells / k_points_in_obb_cells).  The table must never hide a box: points outside the grid, on cell faces, non-finite points,
    a scene far from the origin (float32 bounds are coarse there), boxes that are all the same, boxes with infinite bounds."""
    ctx = f3d.default_context()
    rng = np.random.default_rng(100 + B)

    def check(pts, boxes, dtype=np.float64):
        with np.errstate(all='ignore'):
            inside, cooc = ctx.points_in_obb(pts.astype(dtype), boxes)
            p64 = pts.astype(dtype).astype(np.float64)
            want = np.stack([O.points_in_obb(p64, b[0:3], b[3:12].reshape(3, 3), b[12:15]) for b in boxes], axis=1)
        assert np.array_equal(inside, want)
        m = want.astype(np.float32)
        assert np.array_equal(cooc, (m.T @ m) > 0)
        return want

    def random_boxes(centre, spread, lo_e, hi_e):
        boxes = np.zeros((B, 15))
        boxes[:, 0:3] = centre + rng.uniform(-1, 1, (B, 3)) * spread
        for k in range(B):
            boxes[k, 3:12] = np.linalg.qr(rng.normal(size=(3, 3)))[0].reshape(-1)
        boxes[:, 12:15] = rng.uniform(lo_e, hi_e, (B, 3))
        return boxes

    boxes = random_boxes(np.array([0.0, 0.0, 1.5]), np.array([5.0, 5.0, 1.5]), 0.2, 1.5)
    pts = rng.uniform([-8, -8, -2], [8, 8, 5], (60_000, 3))                       # a third of them outside the grid
    pts[:5] = [[np.nan, 0, 0], [np.inf, 0, 1], [0, -np.inf, 1], [1e300, 1e300, 1e300], [0, 0, np.nan]]
    # points exactly on box centres, box corners and (nearly) on faces
    pts[5:5 + B] = boxes[:, 0:3]
    from Fusion3DSeg.merge_intersecting_bb import obb_corners
    pts[100:108] = obb_corners(boxes[0, 0:3], boxes[0, 3:12].reshape(3, 3), boxes[0, 12:15])
    face = boxes[1, 0:3] + boxes[1, 3:12].reshape(3, 3)[:, 0] * boxes[1, 12] / 2
    pts[110:120] = face + np.outer(np.arange(-5, 5) * 1e-16, boxes[1, 3:12].reshape(3, 3)[:, 0])
    want = check(pts, boxes)
    assert want.any(0).all() and want[5:5 + B].any(1).all()
    check(pts, boxes, np.float32)
    # the same scene 1e6 away from the origin: the float32 bounds and a float32 cloud are coarse (0.06 m), the table has to allow for it
    off = np.array([1.0e6, -2.0e6, 3.0e5])
    far = boxes.copy(); far[:, 0:3] += off
    assert check(pts[120:] + off, far).any()
    assert check(pts[120:] + off, far, np.float32).any()
    # all boxes the same (a grid of no extent along any axis), and boxes without usable bounds among ordinary ones
    same = np.repeat(boxes[:1], B, axis=0)
    assert check(pts, same).all(1).any()
    odd = boxes.copy()
    odd[2, 3:12] = np.array([[1, 2, 3], [2, 4, 6], [0, 0, 1.0]]).reshape(-1)      # singular: infinite bounds
    odd[4, 3] = np.nan
    odd[6, 12:15] = [1e9, 1e9, 1e9]                                               # contains everything finite
    w = check(pts, odd)
    assert w[5:, 6].sum() >= len(pts) - 6


def _rank_labels(rank, world, port, n, out_dir):
    """One rank of the point-sharded step: HIP labels of its shard, masks all-gathered over the process group."""
    import torch
    import torch.distributed as dist
    from f3d import sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sc = synth.scene('C1', n=n)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    ctx = f3d.Context(0)
    v0, v1 = sharding.view_bounds(len(views), rank, world)

    def label_fn(points, masks_full):
        return ctx.project_vote_argmax(points, views, masks_full.numpy(), 133, 0.0, None)

    labels = sharding.sharded_labels(dist, sc['points'], torch.from_numpy(sc['masks'][v0:v1].copy()), label_fn, gather=True)
    np.save(os.path.join(out_dir, f'labels{rank}.npy'), labels.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_label_their_shards_with_the_hip_path(tmp_path):
    """world size 2 (two fresh processes, both on device 0, gloo): unequal shards (n odd), every rank's HIP labels gathered,
    equal to the single-process HIP result and to the oracle."""
    import torch.multiprocessing as mp
    n = 40_001
    mp.spawn(_rank_labels, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    sc, views = _scene(n)
    single = f3d.default_context().project_vote_argmax(sc['points'], views, sc['masks'], 133, 0.0, None)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'labels{r}.npy'), single), r
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, 0.0, None)
    assert np.array_equal(single, want) and (want != 133).mean() > 0.5


def _rank_overlap_coded(rank, world, port, n, nchunks, out_dir):
    """One rank of the step with the CODED exchange: the rank codes its own masks, coded planes are all-gathered (gloo) chunk by
    chunk, f3d_fuse_chunk_coded_dev votes on them; this rank never sees a raw mask of the other rank."""
    import torch
    import torch.distributed as dist
    from f3d import sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    sc = synth.scene('C1', n=n)
    V = 16
    q, t = synth.ring_views(V)
    masks = synth.masks(V, sc['h'], sc['w'], 'block64x40')
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], q, t, sc['max_depth'])
    vc, order = sharding.chunk_layout(V, world, nchunks)
    v0, v1 = sharding.view_bounds(V, rank, world)
    lo, hi = sharding.point_bounds(n, rank, world)
    pts = sc['points'][lo:hi].copy()
    pts[::53] *= 1e31                                               # deferred in every tier: labelled by the reference arithmetic on coded planes
    x = torch.from_numpy(pts).to(dev)
    cls = torch.empty(hi - lo, dtype=torch.int64, device=dev)
    ctx = f3d.Context(0)
    eng = sharding.HipChunkEngine(ctx, x, f3d.F64, hi - lo, torch.from_numpy(views[order]).to(dev), sc['h'], sc['w'], 133, 0.0, None, cls,
                                  flags=f3d.FUSE_SORT)
    shard = torch.from_numpy(masks[v0:v1].copy()).to(dev)
    gathered = torch.empty((V, eng.coded_plane_bytes()), dtype=torch.uint8, device=dev)
    sharding.overlapped_labels_coded(dist, eng, shard, gathered, nchunks)
    ctx.take_device_error(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    labels = sharding.gather_labels(dist, cls.cpu(), n)
    np.save(os.path.join(out_dir, f'coded{rank}.npy'), labels.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_exchange_coded_masks(tmp_path):
    """SURVEY 8(e1) / DESIGN section 5: mask coding sharded by view, coded all-gather, exact tier on coded planes -- labels equal
    the oracle's on the whole cloud."""
    import torch.multiprocessing as mp
    n, V = 30_001, 16
    mp.spawn(_rank_overlap_coded, args=(2, _free_port(), n, 2, str(tmp_path)), nprocs=2, join=True)
    sc = synth.scene('C1', n=n)
    q, t = synth.ring_views(V)
    masks = synth.masks(V, sc['h'], sc['w'], 'block64x40')
    pts = sc['points'].copy()
    for r in range(2):
        lo, hi = n * r // 2, n * (r + 1) // 2
        pts[lo:hi][::53] *= 1e31
    with np.errstate(all='ignore'):
        want = O.project_vote_argmax(pts, sc['K'], q, t, masks, sc['max_depth'], 133, 0.0, None)
    assert (want != 133).mean() > 0.3
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'coded{r}.npy'), want), r


def _rank_overlap(rank, world, port, n, nchunks, out_dir):
    """One rank of the step with the exchange overlapped inside it: chunked all-gather (gloo, host tensors copied to device
    0 as they land) + f3d's view-chunked fused call on this rank's point range."""
    import torch
    import torch.distributed as dist
    from f3d import sharding
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    sc = synth.scene('C1', n=n)
    V = 16
    q, t = synth.ring_views(V)
    masks = synth.masks(V, sc['h'], sc['w'], 'block64x40')
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], q, t, sc['max_depth'])
    vc, order = sharding.chunk_layout(V, world, nchunks)
    v0, v1 = sharding.view_bounds(V, rank, world)
    lo, hi = sharding.point_bounds(n, rank, world)
    x = torch.from_numpy(sc['points'][lo:hi].copy()).to(dev)
    cls = torch.empty(hi - lo, dtype=torch.int64, device=dev)
    ctx = f3d.Context(0)
    eng = sharding.HipChunkEngine(ctx, x, f3d.F64, hi - lo, torch.from_numpy(views[order]).to(dev), sc['h'], sc['w'], 133, 0.0, None, cls,
                                  flags=f3d.FUSE_SORT)
    shard = torch.from_numpy(masks[v0:v1].copy()).to(dev)
    gathered = torch.empty((V,) + tuple(shard.shape[1:]), dtype=torch.uint8, device=dev)
    sharding.overlapped_labels(dist, eng, shard, gathered, nchunks)
    ctx.take_device_error(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    labels = sharding.gather_labels(dist, cls.cpu(), n)
    np.save(os.path.join(out_dir, f'overlap{rank}.npy'), labels.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_overlap_the_mask_exchange_inside_the_step(tmp_path):
    import torch.multiprocessing as mp
    n, V = 30_001, 16
    mp.spawn(_rank_overlap, args=(2, _free_port(), n, 4, str(tmp_path)), nprocs=2, join=True)
    sc = synth.scene('C1', n=n)
    q, t = synth.ring_views(V)
    masks = synth.masks(V, sc['h'], sc['w'], 'block64x40')
    want = O.project_vote_argmax(sc['points'], sc['K'], q, t, masks, sc['max_depth'], 133, 0.0, None)
    assert (want != 133).mean() > 0.3
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'overlap{r}.npy'), want), r


def _merge_scene_hip(B=96, n=60_000, seed=77):
    rng = np.random.default_rng(seed)
    centres = rng.uniform([-3, -3, 0], [3, 3, 2], (B, 3))
    ids = rng.integers(1, B, n).astype(np.int64)
    pts = centres[ids] + rng.normal(size=(n, 3)) * [0.25, 0.15, 0.1]
    info = [{'id': k, 'category_id': 86, 'parent_id': k % 3, 'area': int((ids == k).sum())} for k in range(B)]
    return pts, ids, info


def _rank_merge(rank, world, port, out_dir):
    """One rank of the sharded bbox merge on the HIP path: the default HipCloud backend, this rank's [lo, hi) share of the points in
    every scan, one all_reduce(MAX) per co-occurrence answer, every box fitted on the GPU by every rank (deterministic: no exchange)."""
    import copy
    import json
    import torch.distributed as dist
    from Fusion3DSeg.merge_intersecting_bb import merge_bb
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pts, ids, info = _merge_scene_hip()
        out_info, out_ids = merge_bb(out_dir, copy.deepcopy(info), ids, pts, dist=dist)
        np.save(os.path.join(out_dir, f'merge_ids_{rank}.npy'), out_ids)
        with open(os.path.join(out_dir, f'merge_info_{rank}.json'), 'w') as fp:
            json.dump(out_info, fp)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharded_merge_bb_on_the_hip_path(tmp_path):
    """SURVEY 8(e) / VERDICT r2 (e2): merge_bb(dist=...) with the DEFAULT backend under two ranks (two processes on device 0, gloo):
    HipCloud's sharded scan (device pointer + 24 * lo, hi - lo points) really runs; both ranks' results equal the single-process
    merge_bb bit for bit and the oracle's literal restatement in ids, areas and boxes (corner sets); rank 0 alone writes the files."""
    import copy
    import json
    import torch.multiprocessing as mp
    from Fusion3DSeg.merge_intersecting_bb import merge_bb
    mp.spawn(_rank_merge, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    pts, ids, info = _merge_scene_hip()
    single_info, single_ids = merge_bb(None, copy.deepcopy(info), ids.copy(), pts)
    want_info, want_ids = O.merge_bb(copy.deepcopy(info), ids.copy(), pts)
    assert len(want_info) < len(info) - 5 and np.array_equal(single_ids, want_ids)
    assert [(d['id'], d['area']) for d in single_info] == [(d['id'], d['area']) for d in want_info]
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'merge_ids_{r}.npy'), single_ids), r
        assert json.loads((tmp_path / f'merge_info_{r}.json').read_text()) == json.loads(json.dumps(single_info)), r
    for g, w in zip(single_info, want_info):
        assert ('bbox' in g) == ('bbox' in w)
        if 'bbox' in g:
            d = np.abs(np.array(g['bbox'])[:, None, :] - np.array(w['bbox'])[None, :, :]).max(-1)
            assert (d.min(1) < 1e-8).all() and (d.min(0) < 1e-8).all()
    assert np.array_equal(np.load(tmp_path / 'panoptic_segmentation' / 'ids.npy'), single_ids)
