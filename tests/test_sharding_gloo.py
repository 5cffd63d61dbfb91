"""The N>1 path (one process per GPU, point-sharded cloud, all-gathered masks) on CPU: world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from f3d import sharding, synth
from oracle import np_ref as O


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sc = synth.scene('C1', n=n, mask_kind='iid')
        v0, v1 = sharding.view_bounds(len(sc['masks']), rank, world)
        shard = torch.from_numpy(sc['masks'][v0:v1].copy())            # this rank only ever holds its own views' masks

        def label_fn(points, masks_full):
            assert np.array_equal(masks_full.numpy(), sc['masks'])     # the exchange replicated every mask
            return O.project_vote_argmax(points, sc['K'], sc['wxyzs'], sc['translations'], masks_full.numpy(),
                                         sc['max_depth'], 133, 0.5, [86, 114, 115])

        labels = sharding.sharded_labels(dist, sc['points'], shard, label_fn)
        np.save(os.path.join(out_dir, f'labels_{rank}.npy'), labels.numpy())
    finally:
        dist.destroy_process_group()


def test_point_and_view_bounds():
    for n, w in [(10, 3), (7, 8), (1_000_003, 8), (0, 2)]:
        spans = [sharding.point_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [sharding.view_bounds(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        sharding.view_bounds(10, 0, 4)


def test_two_rank_sharded_labels_equal_single_process(tmp_path):
    n, world = 3001, 2                                                 # odd size: unequal shards
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    sc = synth.scene('C1', n=n, mask_kind='iid')
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                 133, 0.5, [86, 114, 115])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'labels_{r}.npy'), want)


# ------------------------------------------------------------------------------------------------------------
# within-step overlap: chunked all-gather, votes carried from chunk to chunk
# ------------------------------------------------------------------------------------------------------------
class OracleChunkEngine:
    """What HipChunkEngine does on the GPU, with the oracle's per-view votes: the carry is the float64 vote matrix."""

    def __init__(self, points, sc, order, thr, flt):
        self.points, self.sc, self.order, self.thr, self.flt = points, sc, order, thr, flt
        self.votes, self.labels, self.present, self.seen = None, None, None, []

    def presence(self, mask_shard):
        p = torch.zeros(256, dtype=torch.uint8)
        p[torch.unique(mask_shard).long()] = 1
        return p

    def begin(self, present):
        self.present = present.numpy().copy()
        self.votes = np.zeros((len(self.points), 134))

    def chunk(self, v_begin, v_end, gathered):
        sc, sel = self.sc, self.order[v_begin:v_end]
        m = gathered.numpy()[v_begin:v_end]
        assert np.array_equal(m, sc['masks'][sel])                       # the planes that have landed are the chunk's views
        self.seen.append((v_begin, v_end))
        self.votes += O.forward_votes(self.points, sc['K'], sc['wxyzs'][sel], sc['translations'][sel], m, sc['max_depth'], ncols=134)
        if v_end == len(self.order):
            self.labels = O.segment(self.votes, 133, self.thr, self.flt)


class OracleCodedEngine(OracleChunkEngine):
    """The coded exchange on CPU: "coding" = the raw plane behind an 8-byte header that names the book (so that a plane coded
    before begin(), or with another rank's book, is noticed), padded like the real coded planes are larger than raw ones."""

    def coded_plane_bytes(self):
        return self.sc['masks'][0].size + 24

    def code(self, masks, out):
        assert self.present is not None, 'coding needs the book of begin()'
        k = masks.shape[0]
        out[:, :8] = torch.from_numpy(np.frombuffer(np.int64(self.present.sum()).tobytes(), np.uint8).copy())
        out[:, 8:8 + masks[0].numel()] = masks.reshape(k, -1)
        out[:, 8 + masks[0].numel():] = 0

    def chunk_coded(self, v_begin, v_end, gathered):
        g = gathered.numpy()
        assert (np.frombuffer(g[v_begin:v_end, :8].tobytes(), np.int64) == self.present.sum()).all()
        hw = self.sc['masks'][0].size
        planes = torch.from_numpy(g[:, 8:8 + hw].reshape((len(g),) + self.sc['masks'].shape[1:]).copy())
        self.chunk(v_begin, v_end, planes)


def _overlap_coded_worker(rank, world, port, n, nchunks, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sc = _ring_scene(n)
        V = len(sc['masks'])
        v0, v1 = sharding.view_bounds(V, rank, world)
        shard = torch.from_numpy(sc['masks'][v0:v1].copy())
        vc, order = sharding.chunk_layout(V, world, nchunks)
        lo, hi = sharding.point_bounds(n, rank, world)
        eng = OracleCodedEngine(sc['points'][lo:hi], sc, order, 0.0, None)
        gathered = torch.full((V, eng.coded_plane_bytes()), 255, dtype=torch.uint8)
        sharding.overlapped_labels_coded(dist, eng, shard, gathered, nchunks)
        assert eng.seen == [(c * world * vc, (c + 1) * world * vc) for c in range(nchunks)]
        assert np.array_equal(np.flatnonzero(eng.present), np.unique(sc['masks']))     # the union over BOTH ranks' masks
        labels = sharding.gather_labels(dist, torch.from_numpy(eng.labels), n)
        np.save(os.path.join(out_dir, f'coded_{rank}.npy'), labels.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('nchunks', [1, 2])
def test_two_rank_coded_exchange_equals_single_process(tmp_path, nchunks):
    """overlapped_labels_coded on CPU (gloo, world 2): the book is agreed before any plane is coded, each rank codes only its own
    planes, the coded planes land in chunk order, labels equal the single-process oracle."""
    n, world = 2001, 2
    mp.spawn(_overlap_coded_worker, args=(world, _free_port(), n, nchunks, str(tmp_path)), nprocs=world, join=True)
    sc = _ring_scene(n)
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, 0.0, None)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'coded_{r}.npy'), want)


def _overlap_worker(rank, world, port, n, nchunks, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sc = _ring_scene(n)
        V = len(sc['masks'])
        v0, v1 = sharding.view_bounds(V, rank, world)
        shard = torch.from_numpy(sc['masks'][v0:v1].copy())
        vc, order = sharding.chunk_layout(V, world, nchunks)
        lo, hi = sharding.point_bounds(n, rank, world)
        eng = OracleChunkEngine(sc['points'][lo:hi], sc, order, 0.0, None)
        gathered = torch.full((V,) + shard.shape[1:], 255, dtype=torch.uint8)
        sharding.overlapped_labels(dist, eng, shard, gathered, nchunks)
        assert eng.seen == [(c * world * vc, (c + 1) * world * vc) for c in range(nchunks)]
        assert np.array_equal(np.flatnonzero(eng.present), np.unique(sc['masks']))     # the union over BOTH ranks' masks
        labels = sharding.gather_labels(dist, torch.from_numpy(eng.labels), n)
        np.save(os.path.join(out_dir, f'overlap_{rank}.npy'), labels.numpy())
    finally:
        dist.destroy_process_group()


def _ring_scene(n, V=8):
    sc = synth.scene('C1', n=n)
    q, t = synth.ring_views(V)
    sc['wxyzs'], sc['translations'] = q, t
    sc['masks'] = synth.masks(V, sc['h'], sc['w'], 'block64')
    sc['masks'][V // 2:][sc['masks'][V // 2:] == sc['masks'][0, 0, 0]] = 77       # a label only the second rank's masks hold
    return sc


def test_chunk_layout():
    vc, order = sharding.chunk_layout(16, 2, 4)
    assert vc == 2 and order.tolist() == [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
    assert sharding.chunk_layout(64, 8, 1)[1].tolist() == list(range(64))
    with pytest.raises(ValueError):
        sharding.chunk_layout(64, 8, 3)


@pytest.mark.parametrize('nchunks', [1, 2, 4])
def test_two_rank_overlapped_step_equals_single_process(tmp_path, nchunks):
    n, world = 2001, 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), n, nchunks, str(tmp_path)), nprocs=world, join=True)
    sc = _ring_scene(n)
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, 0.0, None)
    assert (want != 133).mean() > 0.3
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'overlap_{r}.npy'), want)


# ------------------------------------------------------------------------------------------------------------
# bbox merge: scans sharded by point range, one all_reduce(MAX) per co-occurrence answer, initial
# box fits dealt out by instance and all-gathered, control flow replicated on every rank
# ------------------------------------------------------------------------------------------------------------
class NumpyCloud:
    """Stand-in for the HIP cloud on a machine without a GPU: same interface, the oracle's in-box test, this rank's points only."""

    def __init__(self, pts, share):
        self.pts, (self.lo, self.hi) = pts, share

    def group(self, ids, nids):
        order = np.argsort(ids, kind='stable').astype(np.int32)
        return order, np.searchsorted(ids[order], np.arange(nids + 2)).astype(np.int64)

    def cooccurrence(self, packed):
        mine = self.pts[self.lo:self.hi]
        m = np.stack([O.points_in_obb(mine, b[0:3], b[3:12].reshape(3, 3), b[12:15]) for b in packed]).astype(np.float32)
        return ((m @ m.T) > 0).astype(np.uint8)


def _merge_scene(seed=9):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0, 3.0, (16, 3))
    pts = np.vstack([c + rng.normal(size=(250, 3)) * [0.5, 0.3, 0.1] for c in centres])
    ids = np.repeat(np.arange(16), 250).astype(np.int64)
    perm = rng.permutation(len(pts))                                   # members of an instance spread over both shards
    pts, ids = pts[perm], ids[perm]
    ids[ids == 15] = 14                                                # one instance with no points at all
    info = [{'id': k, 'category_id': 86 + (k % 3), 'parent_id': k % 2, 'area': int((ids == k).sum())} for k in range(16)]
    return pts, ids, info


def _merge_worker(rank, world, port, out_dir):
    import copy
    import json
    from Fusion3DSeg.merge_intersecting_bb import merge_bb
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pts, ids, info = _merge_scene()
        out_info, out_ids = merge_bb(out_dir, copy.deepcopy(info), ids, pts, box_fn=O.obb_from_points, dist=dist, backend=NumpyCloud)
        np.save(os.path.join(out_dir, f'ids_{rank}.npy'), out_ids)
        with open(os.path.join(out_dir, f'info_{rank}.json'), 'w') as fp:
            json.dump(out_info, fp)
    finally:
        dist.destroy_process_group()


def test_sharded_cooccurrence_is_an_or_over_ranks():
    class FakeDist:                                                    # one "rank": the reduction is the identity, the dtype contract holds
        ReduceOp = dist.ReduceOp

        @staticmethod
        def get_backend():
            return 'gloo'

        @staticmethod
        def all_reduce(t, op=None):
            return None
    got = sharding.sharded_cooccurrence(FakeDist, np.array([[1, 0], [0, 1]], bool))
    assert got.dtype == np.uint8 and got.tolist() == [[1, 0], [0, 1]]


def test_two_rank_sharded_merge_bb_equals_single_process(tmp_path):
    """world size 2, gloo: every rank scans half of the points; ids, areas, surviving entries and boxes equal the oracle's
    single-process literal restatement (reference merge_intersecting_bb.py:103-137); rank 0 alone writes the files."""
    import copy
    import json
    world = 2
    mp.spawn(_merge_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pts, ids, info = _merge_scene()
    want_info, want_ids = O.merge_bb(copy.deepcopy(info), ids.copy(), pts)
    assert len(want_info) < len(info)                                  # something merged
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f'ids_{r}.npy'), want_ids), r
        got = json.loads((tmp_path / f'info_{r}.json').read_text())
        assert [d['id'] for d in got] == [d['id'] for d in want_info] and [d['area'] for d in got] == [d['area'] for d in want_info]
        for g, w in zip(got, want_info):
            assert ('bbox' in g) == ('bbox' in w)
            if 'bbox' in w:
                assert np.array_equal(np.array(g['bbox']), np.array(w['bbox']))
    assert np.array_equal(np.load(tmp_path / 'panoptic_segmentation' / 'ids.npy'), want_ids)
