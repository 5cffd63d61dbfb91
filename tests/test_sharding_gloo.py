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
