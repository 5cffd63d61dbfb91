"""Multi-GPU layout of the hot path: one process per GPU, the cloud sharded by contiguous point ranges.

Every point's label depends only on its own xyz and on the (replicated) views, so there is no reduction.  The one
exchange step is making all V masks available on every rank: each rank owns the masks of the views it produced
(its share of the 2D stage), and one all-gather over RCCL/xGMI (``torch.distributed`` backend "nccl"; "gloo" in the
CPU tests) replicates them.  Labels stay sharded, or are gathered with ``gather_labels``.
"""
import numpy as np


def point_bounds(n, rank, world):
    """Contiguous range [lo, hi) of the n points owned by `rank` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def view_bounds(nviews, rank, world):
    """Views whose masks `rank` produces; equal shares are required by the single all-gather."""
    if nviews % world:
        raise ValueError(f'{nviews} views cannot be split evenly over {world} ranks (pad the view list)')
    per = nviews // world
    return rank * per, (rank + 1) * per


def all_gather_masks(dist, shard, out=None):
    """shard: uint8 tensor [V/world, H, W] on this rank -> uint8 [V, H, W] on every rank (one collective)."""
    import torch
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((shard.shape[0] * world,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    try:
        dist.all_gather_into_tensor(out.view(-1), shard.contiguous().view(-1))
    except (RuntimeError, NotImplementedError):                 # a backend without the flat form
        parts = list(out.view(world, -1).unbind(0))
        dist.all_gather(parts, shard.contiguous().view(-1))
    return out


def gather_labels(dist, local, n_total):
    """int64 labels of this rank's point range -> the full [n_total] vector on every rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    longest = max(point_bounds(n_total, r, world)[1] - point_bounds(n_total, r, world)[0] for r in range(world))
    pad = torch.full((longest,), -1, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty(n_total, dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = point_bounds(n_total, r, world)
        out[lo:hi] = parts[r][:hi - lo]
    return out


def sharded_labels(dist, points, mask_shard, label_fn, gather=True):
    """The N-rank step: all-gather the masks, label this rank's contiguous share of `points` with
    ``label_fn(points_shard, masks_full) -> int64 labels``, optionally gather the labels."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    masks = all_gather_masks(dist, mask_shard)
    lo, hi = point_bounds(len(points), rank, world)
    local = label_fn(points[lo:hi], masks)
    if not gather:
        return local
    local_t = torch.as_tensor(np.ascontiguousarray(local)) if not isinstance(local, torch.Tensor) else local
    return gather_labels(dist, local_t, len(points))


def sharded_cooccurrence(dist, local):
    """The bbox-merge exchange step: every rank has scanned ITS share of the points against the same boxes;
    `local` (uint8 / bool array of any shape: a B x B co-occurrence matrix or one row of it) says which box pairs share a point
    of that share.  One all_reduce(MAX) -- "some rank saw a common point" -- gives every rank the answer for the whole cloud."""
    import torch
    arr = np.ascontiguousarray(local, dtype=np.uint8)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.from_numpy(arr.astype(np.int32)).to(dev)          # gloo has no MAX for uint8
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy().astype(np.uint8)
