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


# ---- within-step overlap: the mask all-gather split into view chunks, chunk c+1 in flight while chunk c votes --------------
def chunk_layout(nviews, world, nchunks):
    """(views per rank and chunk, view index of every plane of the chunked gather buffer).

    Rank r owns the masks of views [r*per, (r+1)*per) and sends them in `nchunks` equal slices; collective c gathers slice c of
    every rank, so the buffer holds chunk after chunk, inside a chunk rank after rank.  Votes are a sum over views, hence the
    labels do not depend on this order -- only the view records have to be handed over in the same one."""
    if nviews % world or (nviews // world) % nchunks:
        raise ValueError(f'{nviews} views do not split into {world} ranks x {nchunks} equal chunks')
    per = nviews // world
    vc = per // nchunks
    order = np.array([r * per + c * vc + k for c in range(nchunks) for r in range(world) for k in range(vc)], dtype=np.int64)
    return vc, order


class HipChunkEngine:
    """The fused path of one rank behind `overlapped_labels`: device pointers into f3d's view-chunked entry points."""

    def __init__(self, ctx, xyz, dtype, n, views_in_chunk_order, h, w, nclasses, threshold, filter_classes, classes, flags=0, perm_ptr=None):
        import torch
        self.ctx, self.xyz, self.dtype, self.n, self.views = ctx, xyz, dtype, n, views_in_chunk_order
        self.h, self.w, self.nclasses, self.threshold, self.flt = h, w, nclasses, threshold, filter_classes
        self.classes, self.flags, self.perm_ptr = classes, flags, perm_ptr
        self.present = torch.empty(256, dtype=torch.uint8, device=xyz.device)

    def _stream(self):
        import torch
        return torch.cuda.current_stream(self.xyz.device).cuda_stream

    def presence(self, mask_shard):
        self.ctx.mask_presence_dev(mask_shard.data_ptr(), mask_shard.shape[0], self.h, self.w, self.present.data_ptr(), self._stream())
        return self.present

    def begin(self, present):
        self.ctx.fuse_chunked_begin_dev(None if present is None else present.data_ptr(), self.n, len(self.views), self.h, self.w,
                                        self.nclasses, self.flt, self._stream())

    def chunk(self, v_begin, v_end, masks):
        self.ctx.fuse_chunk_dev(self.xyz.data_ptr(), self.dtype, self.n, self.views.data_ptr(), len(self.views), v_begin, v_end,
                                masks.data_ptr(), self.h, self.w, self.nclasses, self.threshold, self.flt, self.classes.data_ptr(),
                                self._stream(), flags=self.flags, perm_ptr=self.perm_ptr)

    # the coded exchange: this rank codes its own planes (the book of begin() is the same on every rank), coded planes travel
    def coded_plane_bytes(self):
        return self.ctx.coded_plane_bytes(self.h, self.w)

    def code(self, masks, out):
        """masks uint8 [k, H, W] -> out uint8 [k, coded_plane_bytes] (enqueue only)."""
        self.ctx.code_planes_dev(masks.data_ptr(), masks.shape[0], self.h, self.w, out.data_ptr(), self._stream())

    def chunk_coded(self, v_begin, v_end, coded):
        self.ctx.fuse_chunk_coded_dev(self.xyz.data_ptr(), self.dtype, self.n, self.views.data_ptr(), len(self.views), v_begin, v_end,
                                      coded.data_ptr(), self.h, self.w, self.nclasses, self.threshold, self.flt, self.classes.data_ptr(),
                                      self._stream(), flags=self.flags, perm_ptr=self.perm_ptr)


def overlapped_labels(dist, engine, mask_shard, gathered, nchunks):
    """One N-rank step with the exchange overlapped INSIDE the step.

    mask_shard: uint8 [V/world, H, W] (this rank's masks);  gathered: uint8 [V, H, W] buffer in `chunk_layout` order.
    1. the labels present in the local masks, all-reduced (MAX) -> the same vote-bin code book on every rank;
    2. `nchunks` asynchronous all-gathers, all enqueued at once (the backend runs them in order on its own stream);
    3. as soon as chunk c has landed, every local point votes on its views (engine.chunk) while chunk c+1.. are on the wire.
    The per-point vote state between chunks lives in the engine (f3d: packed 8-bit bins in HBM, see f3d_fuse_chunk_dev)."""
    import torch
    world = dist.get_world_size()
    per = mask_shard.shape[0]
    if per % nchunks:
        raise ValueError(f'{per} masks per rank do not split into {nchunks} chunks')
    vc = per // nchunks
    plane = mask_shard[0].numel()
    present = engine.presence(mask_shard)
    pres_work = None
    if present is not None:
        pres32 = present.to(torch.int32)                            # gloo has no MAX for uint8
        pres_work = dist.all_reduce(pres32, op=dist.ReduceOp.MAX, async_op=True)
    flat, shard_flat = gathered.view(-1), mask_shard.contiguous().view(-1)
    works = []
    for c in range(nchunks):
        out = flat[c * world * vc * plane:(c + 1) * world * vc * plane]
        src = shard_flat[c * vc * plane:(c + 1) * vc * plane]
        try:
            works.append(dist.all_gather_into_tensor(out, src, async_op=True))
        except (RuntimeError, NotImplementedError):                 # a backend without the flat form
            works.append(dist.all_gather(list(out.view(world, -1).unbind(0)), src, async_op=True))
    if pres_work is not None:
        pres_work.wait()                                            # device backends: the current stream waits, the host does not
        present.copy_(pres32.to(torch.uint8))
    engine.begin(present)
    for c in range(nchunks):
        works[c].wait()
        engine.chunk(c * world * vc, (c + 1) * world * vc, gathered)


def overlapped_labels_coded(dist, engine, mask_shard, gathered_coded, nchunks):
    """`overlapped_labels` with the mask CODING sharded as well: every rank codes only its own V / world masks and the ranks
    all-gather coded planes (3 % more bytes on the wire, 1 / world of the coding work per rank, and no rank ever holds
    a raw mask of another rank -- the engine's last tier works on codes).

    gathered_coded: uint8 [V, engine.coded_plane_bytes()] buffer, planes in `chunk_layout` order.
    1. labels present locally, all-reduced (MAX) -> the same code book on every rank (engine.begin);
    2. per chunk: code the local slice, enqueue its all-gather (asynchronous; the coding of slice c+1 overlaps the transfer of c);
    3. as soon as chunk c has landed the local points vote on it (engine.chunk_coded)."""
    import torch
    world = dist.get_world_size()
    per = mask_shard.shape[0]
    if per % nchunks:
        raise ValueError(f'{per} masks per rank do not split into {nchunks} chunks')
    vc = per // nchunks
    plane = engine.coded_plane_bytes()
    present = engine.presence(mask_shard)
    if present is not None:
        pres32 = present.to(torch.int32)                            # gloo has no MAX for uint8
        dist.all_reduce(pres32, op=dist.ReduceOp.MAX)
        present.copy_(pres32.to(torch.uint8))
    engine.begin(present)
    mine = torch.empty((per, plane), dtype=torch.uint8, device=mask_shard.device)
    flat = gathered_coded.view(-1)
    works = []
    for c in range(nchunks):
        engine.code(mask_shard[c * vc:(c + 1) * vc], mine[c * vc:(c + 1) * vc])
        if mask_shard.is_cuda and dist.get_backend() != 'nccl':
            torch.cuda.current_stream(mask_shard.device).synchronize()   # a host-side backend reads the buffer: the coding must have finished
        out = flat[c * world * vc * plane:(c + 1) * world * vc * plane]
        src = mine[c * vc:(c + 1) * vc].view(-1)
        try:
            works.append(dist.all_gather_into_tensor(out, src, async_op=True))
        except (RuntimeError, NotImplementedError):                 # a backend without the flat form
            works.append(dist.all_gather(list(out.view(world, -1).unbind(0)), src, async_op=True))
    for c in range(nchunks):
        works[c].wait()
        engine.chunk_coded(c * world * vc, (c + 1) * world * vc, gathered_coded)
