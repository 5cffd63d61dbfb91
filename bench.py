#!/usr/bin/env python3
"""Headline benchmark: labelled points / second of the fused Fusion3DSeg hot path on MI355X.

One step = one pass of the hot path over one batch: every point of the rank's cloud shard is
projected into all V views, sampled, voted and segmented (f3d_project_vote_argmax_dev), with the
inputs already resident in HBM.  Workload at N=1: BASELINE.json config C3 (10M points x 64 views,
1024x1024 masks).  With N>1 ranks (torchrun, one per GPU) every rank owns a 10M-point shard of an
N x 10M cloud and the V/N views whose masks it "produced"; each step all-gathers the masks over
RCCL (the path's one exchange step) and then fuses its shard -- weak scaling.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--filter] [--masks iid|block64]

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
PKG = ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'
for _p in (str(ROOT), str(PKG)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP64_VALU_PEAK_TFLOPS = 78.6     # vector fp64 (FMA = 2 flop)
FLOP_PER_POINT_VIEW = 82         # algorithmic fp64 flop of frustum + projection


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--points', type=int, default=10_000_000, help='points per GPU (C3: 10M)')
    ap.add_argument('--views', type=int, default=64)
    ap.add_argument('--size', type=int, default=1024, help='mask width = height')
    ap.add_argument('--masks', default='block64', choices=['block64', 'iid'])
    ap.add_argument('--filter', action='store_true', help='segment with the reference default filter_classes=[86,114,115]')
    ap.add_argument('--f32', action='store_true', help='store xyz as float32 (12 B/point) instead of float64')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the streaming-kernel measurements')
    ap.add_argument('--cpu-sample', type=int, default=5_000_000, help='points of the CPU-baseline slice (all views)')
    ap.add_argument('--sorted', action='store_true', help='experiment: hand the cloud over already in grid-cell order (host sort)')
    ap.add_argument('--no-sort', action='store_true', help='do not cell-sort the cloud inside the step')
    ap.add_argument('--prepared', action='store_true', help='cell-sort once outside the timed loop and keep the sorted cloud resident')
    ap.add_argument('--overlap-sort', action='store_true', help='sort the cloud of step i+1 on a second stream while step i is fused (measured: no '
                    'gain, 1.43 vs 1.39 ms -- the fused kernel holds every wave slot of the chip, the sort kernels queue behind it)')
    return ap.parse_args()


def algorithmic_bytes(n, v, h, w, xyz_bytes):
    """xyz read once + masks read once + int64 classes written + view records."""
    return xyz_bytes * n + v * h * w + 8 * n + 704 * v


def time_kernel(torch, fn, iters, stream):
    """Average device time of fn() over `iters` launches, HIP events on the launch stream."""
    fn(); stream.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(iters):
        fn()
    b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def cpu_baseline(sample, views_n, size, mask_kind, filter_classes):
    """The NumPy port of the reference path (oracle/np_ref.py) on a bounded slice of the same workload
    (about 10-30 s of single-process CPU work), processed in 250k-point pieces to bound memory."""
    from f3d import synth
    from oracle import np_ref as O
    K = np.array([[800., 0, size / 2], [0, 800., size / 2], [0, 0, 1]])
    q, t = synth.ring_views(views_n)
    pts = synth.cloud(sample)
    masks = synth.masks(views_n, size, size, mask_kind)
    out = np.empty(sample, np.int64)
    t0 = time.perf_counter()
    for s0 in range(0, sample, 250_000):
        out[s0:s0 + 250_000] = O.project_vote_argmax(pts[s0:s0 + 250_000], K, q, t, masks, 10.0, 133, 0.5, filter_classes)
    dt = time.perf_counter() - t0
    return dict(value=round(sample / dt, 1), unit='points/s', cores=1, kind='port',
                sample=f'{sample} points x {views_n} views in {dt:.1f} s, one NumPy process ({os.cpu_count()} host cpus visible, '
                       f'elementwise NumPy only -> 1 core); the path is linear in N'), out, pts, masks


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import f3d
    from f3d import sharding, synth
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback on the product path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    use_dist = world > 1 or os.environ.get('F3D_BENCH_FORCE_DIST') == '1'     # the env var rehearses the RCCL path on one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')       # only matters for the one-rank rehearsal; torchrun sets all of these
        os.environ.setdefault('RANK', str(rank)); os.environ.setdefault('WORLD_SIZE', str(world))
        dist.init_process_group('nccl', device_id=dev)
    ctx = f3d.Context(local)

    n, V, S = args.points, args.views, args.size
    flt = [86, 114, 115] if args.filter else None
    K = np.array([[800., 0, S / 2], [0, 800., S / 2], [0, 0, 1]])
    q, t = synth.ring_views(V)
    views_np = f3d.views_build(K, S, S, q, t, 10.0)
    xyz_np = synth.cloud(n, dtype=np.float32 if args.f32 else np.float64, shard=rank)
    if args.sorted:
        cell = np.floor((xyz_np.astype(np.float64) - np.array([-5, -5, 0])) / 0.25).astype(np.int64)
        xyz_np = xyz_np[np.argsort((cell[:, 0] * 64 + cell[:, 1]) * 16 + cell[:, 2], kind='stable')]
    xyz = torch.from_numpy(xyz_np).to(dev)
    del xyz_np
    masks_np = synth.masks(V, S, S, args.masks)
    views = torch.from_numpy(views_np).to(dev)
    masks_full = torch.empty((V, S, S), dtype=torch.uint8, device=dev)
    if use_dist:
        v0, v1 = sharding.view_bounds(V, rank, world)                # this rank "produced" (owns) the masks of views [v0, v1)
        masks_shard = torch.from_numpy(masks_np[v0:v1]).to(dev)
    else:
        masks_full.copy_(torch.from_numpy(masks_np))
    classes = torch.empty(n, dtype=torch.int64, device=dev)
    dtype = f3d.F32 if args.f32 else f3d.F64
    stream = torch.cuda.Stream(dev)          # a real (non-null) HIP stream: kernels, RCCL and the timing events all use it
    torch.cuda.set_stream(stream)

    flags = 0
    perm_ptr = None
    layout = 'caller order, cell-sorted inside every step (F3D_FUSE_SORT)'
    if args.prepared:
        xyz_sorted = torch.empty_like(xyz)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.cloud_sort_cells_dev(xyz.data_ptr(), dtype, n, xyz_sorted.data_ptr(), perm.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        xyz, perm_ptr = xyz_sorted, perm.data_ptr()
        layout = 'prepared: cell-sorted once outside the timed region, sorted copy + permutation resident'
    elif args.no_sort or args.sorted:
        layout = 'caller order, no sort' + (' (cloud handed over pre-sorted by the host)' if args.sorted else '')
    else:
        flags |= f3d.FUSE_SORT
    # The sort of a cloud and the fused kernel of the previous cloud are independent: in a stream of clouds the sort of
    # step i+1 runs on a second HIP stream (its own context = its own scratch) while step i is fused.  Every step still does
    # all of its work (sort, mask coding, fused kernel, exact kernel); the steps overlap in time.
    overlap = bool(flags & f3d.FUSE_SORT) and args.overlap_sort
    if overlap:
        layout = 'caller order; the cell sort of step i+1 runs on a second stream while step i is fused'
        ctx_sort = f3d.Context(local)
        sort_stream = torch.cuda.Stream(dev)
        perm_buf = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
        ev_sorted = [torch.cuda.Event() for _ in range(2)]
        ev_fused = [torch.cuda.Event() for _ in range(2)]
        sort_issued = set()

        def issue_sort(i):
            if i in sort_issued:
                return
            sort_issued.add(i)
            if i >= 2:
                sort_stream.wait_event(ev_fused[i % 2])          # fuse(i-2) was the last reader of this permutation buffer
            ctx_sort.cloud_sort_cells_dev(xyz.data_ptr(), dtype, n, None, perm_buf[i % 2].data_ptr(), sort_stream.cuda_stream)
            ev_sorted[i % 2].record(sort_stream)

    fuse_no = [0]

    def fuse(masks_t=None):
        m = masks_full if masks_t is None else masks_t
        if not overlap:
            ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, m.data_ptr(), S, S,
                                        133, 0.5, flt, classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
            return
        i = fuse_no[0]
        issue_sort(i)
        issue_sort(i + 1)
        stream.wait_event(ev_sorted[i % 2])
        ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, m.data_ptr(), S, S, 133, 0.5, flt,
                                    classes.data_ptr(), None, stream.cuda_stream, flags=f3d.FUSE_GATHER, perm_ptr=perm_buf[i % 2].data_ptr())
        ev_fused[i % 2].record(stream)
        fuse_no[0] = i + 1

    # N > 1: the mask all-gather of step i+1 (RCCL, its own stream) overlaps the kernels of step i (double-buffered masks).
    # Issued BEFORE fuse(i): the collective then only waits for fuse(i-1), the last reader of the buffer it overwrites.
    mask_buf = [masks_full, torch.empty_like(masks_full)] if use_dist else None
    pending = {}
    step_no = [0]

    def issue_gather(i):
        out = mask_buf[i % 2]
        pending[i] = dist.all_gather_into_tensor(out.view(-1), masks_shard.view(-1), async_op=True)

    def step():
        if not use_dist:
            fuse()
            return
        i = step_no[0]
        if i not in pending:
            issue_gather(i)
        issue_gather(i + 1)
        pending.pop(i).wait()                     # the compute stream waits for masks(i); the host does not block
        fuse(mask_buf[i % 2])
        step_no[0] = i + 1

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ctx.take_device_error(stream.cuda_stream)
    deferred = ctx.fuse_deferred(stream.cuda_stream)        # (to the float64 tier, to the exact kernel) in the last step

    # dominant kernel alone (k_fuse), HIP events on its launch stream, same resident inputs: with the in-step
    # sort the step is [sort kernels][k_fuse reading xyz through perm]; both parts are timed on their own.
    k_iters = max(3, min(args.steps, 10))
    t_sort = 0.0
    if flags & f3d.FUSE_SORT:
        perm_t = torch.empty(n, dtype=torch.int32, device=dev)
        t_sort = time_kernel(torch, lambda: ctx.cloud_sort_cells_dev(xyz.data_ptr(), dtype, n, None, perm_t.data_ptr(),
                                                                     stream.cuda_stream), k_iters, stream)
        t_kernel = time_kernel(torch, lambda: ctx.project_vote_argmax_dev(
            xyz.data_ptr(), dtype, n, views.data_ptr(), V, masks_full.data_ptr(), S, S, 133, 0.5, flt, classes.data_ptr(), None,
            stream.cuda_stream, flags=f3d.FUSE_GATHER, perm_ptr=perm_t.data_ptr()), k_iters, stream)
    else:
        t_kernel = time_kernel(torch, fuse, k_iters, stream)
    xyz_b = 12 if args.f32 else 24
    abytes = algorithmic_bytes(n, V, S, S, xyz_b)
    achieved = abytes / t_kernel / 1e9
    traffic, traffic_src = None, None
    pmc = ROOT / 'profiles' / 'r01_pmc_by_kernel.csv'      # PMC passes of this same command (rocprofv3 --pmc, separate runs)
    if pmc.is_file() and n == 10_000_000 and V == 64 and S == 1024 and not args.f32:
        vals = {}
        for line in pmc.read_text().splitlines()[1:]:
            k, c, v = line.split(',')
            if k == 'k_fuse':
                vals[c] = float(v)
        if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
            traffic = int((vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)
            traffic_src = ('profiles/r01_pmc_by_kernel.csv: (FETCH_SIZE + WRITE_SIZE) KB per k_fuse launch, raw (the x2 FETCH_SIZE '
                           'correction is calibrated for wide streams, not for 1-byte gathers; L2 misses incl. Infinity-Cache hits)')
    roofline = dict(bound='hbm', achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                    kernel='k_fuse', kernel_ms=round(t_kernel * 1e3, 4), sort_ms=round(t_sort * 1e3, 4),
                    algorithmic_bytes=abytes,
                    valu_frac=round(FLOP_PER_POINT_VIEW * n * V / t_kernel / (FP64_VALU_PEAK_TFLOPS * 1e12), 4),
                    note='the fused V-view kernel is fp64-VALU bound, not HBM bound (SURVEY 8(d): HBM-fraction ceiling ~7% at '
                         'V=64); valu_frac = 82 reference flop x N x V / t / 78.6 TF (the kernel executes fewer flop than the '
                         'reference formulation thanks to tile culling and the fast projection)')

    out = None
    if rank == 0:
        total_points = n * world
        out = dict(metric='labelled points/sec (10M pts x 64 views) at 1/2/4/8 GPU; % HBM roofline',
                   value=round(total_points * args.steps / elapsed, 1), unit='points/s', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(workload=f'C3: {n} points/GPU x {V} ring views, {S}x{S} {args.masks} uint8 masks, '
                                        f'nclasses=133, threshold=0.5, filter_classes={flt}; fused project->sample->vote->segment',
                               points_per_gpu=n, views=V, mask_hw=[S, S], xyz_storage='f32' if args.f32 else 'f64', cloud_layout=layout,
                               deferred_points=dict(to_float64_tier=deferred[0], to_exact_kernel=deferred[1]),
                               exchange='none' if not use_dist else f'RCCL all_gather of {V // world} masks/rank per step, double-buffered and overlapped with the previous step'),
                   roofline=roofline)

    # secondary, HBM-streaming kernels of the same path (not part of `value`)
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        uv = torch.empty((2, n), dtype=torch.int32, device=dev)
        ins = torch.empty(n, dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.project_view_dev(xyz.data_ptr(), dtype, n, views_np[0], uv.data_ptr(), ins.data_ptr(),
                                                             stream.cuda_stream), 10, stream)
        b = (xyz_b + 8 + 1) * n
        extras['project_view (a2+a4, 1 view)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                        bytes_per_point=xyz_b + 9)
        del uv, ins
        ns = min(n, 4_000_000)
        votes = torch.zeros((ns, 134), dtype=torch.float64, device=dev)
        votes.view(-1)[::7] = 3.0
        cls2 = torch.empty(ns, dtype=torch.int64, device=dev)
        tk = time_kernel(torch, lambda: ctx.segment_votes_dev(votes.data_ptr(), ns, 134, 133, 0.5, None, cls2.data_ptr(), stream.cuda_stream), 10, stream)
        b = (134 * 8 + 8) * ns
        extras['segment_votes (a8)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                             bytes_per_point=1080, points=ns)
        del votes, cls2
        # a7: one frame of the uv2pt scatter vote (1024x1024 lookup, ~70 % valid, 4M points)
        hw = S * S
        rng = np.random.default_rng(7)
        lut_np = rng.integers(0, ns, hw).astype(np.int32); lut_np[rng.random(hw) < 0.3] = -1
        lut = torch.from_numpy(lut_np).to(dev)
        votes = torch.zeros((ns, 134), dtype=torch.float64, device=dev)
        m0 = masks_full[0].reshape(-1)
        tk = time_kernel(torch, lambda: ctx.vote_uv2pt_dev(lut.data_ptr(), m0.data_ptr(), hw, votes.data_ptr(), ns, 134, stream.cuda_stream), 10, stream)
        nvalid = int((lut_np != -1).sum())
        b = 5 * hw + 16 * nvalid
        extras['vote_uv2pt (a7, 1 frame 1024x1024)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                            note='5 B/pixel streamed + 8 B read + 8 B write per vote at random rows (scatter: line-granular traffic is ~8x that)')
        del votes, lut
        # a9: logits -> mask for one 133 x 1024 x 1024 image
        sem = torch.randn((133, S, S), dtype=torch.float32, device=dev)
        mk = torch.empty((S, S), dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.sem_logits_to_mask_dev(sem.data_ptr(), 133, hw, 0.017, 133, mk.data_ptr(), stream.cuda_stream), 10, stream)
        b = (133 * 4 + 1) * hw
        extras['sem_logits_to_mask (a9, 133x1024x1024)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                                  bytes_per_pixel=533)
        del sem, mk
        # a1 / a4 single-purpose streaming kernels
        ins = torch.empty(n, dtype=torch.uint8, device=dev)
        F = f3d.view_fields(views_np)
        pp, pn = np.ascontiguousarray(F['plane_pt'][0]), np.ascontiguousarray(F['plane_n'][0])
        if not args.f32:
            tk = time_kernel(torch, lambda: ctx._check(ctx._lib.f3d_inside_polyhedra_dev(ctx._h, xyz.data_ptr(), dtype, n, pp.ctypes.data, pn.ctypes.data, 5,
                                                                                       ins.data_ptr(), stream.cuda_stream)), 10, stream)
            b = (xyz_b + 1) * n
            extras['inside_polyhedra (a4, 5 planes)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                             bytes_per_point=xyz_b + 1)
        # (f)#3: one 1024x1024 16-bit depth frame -> world points
        dep = torch.randint(0, 6000, (S, S), device=dev, dtype=torch.int16)       # < 32768: the same bits as uint16
        wpts = torch.empty((S * S, 3), dtype=torch.float64, device=dev)
        Kd = np.ascontiguousarray(K, dtype=np.float64); qd = np.array([0.5, 0.5, -0.5, 0.5]); td = np.array([0.25, -0.5, 1.0])
        tk = time_kernel(torch, lambda: ctx._check(ctx._lib.f3d_unproject_depth_dev(ctx._h, dep.data_ptr(), 2, S, S, Kd.ctypes.data, 1000.0, qd.ctypes.data,
                                                                                    td.ctypes.data, wpts.data_ptr(), stream.cuda_stream)), 10, stream)
        b = 26 * S * S
        extras['unproject_depth ((f)#3, 1024x1024 u16 frame)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                                       bytes_per_pixel=26)
        del dep, wpts
        # a10/a11: all points x 64 oriented boxes, membership co-occurrence only
        boxes = np.zeros((64, 15)); boxes[:, 0:3] = rng.uniform([-5, -5, 0], [5, 5, 3], (64, 3))
        boxes[:, 3:12] = np.eye(3).reshape(-1); boxes[:, 12:15] = 0.8
        cooc = torch.zeros((64, 64), dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.points_in_obb_dev(xyz.data_ptr(), dtype, n, boxes, None, cooc.data_ptr(), stream.cuda_stream), 5, stream)
        extras['points_in_obb (a10/a11, 64 boxes)'] = dict(ms=round(tk * 1e3, 4), point_box_tests_per_s=round(64 * n / tk, 1),
                                                           GBps=round(xyz_b * n / tk / 1e9, 1), hbm_frac=round(xyz_b * n / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                           note='fp64-VALU bound beyond ~8 boxes (27 flop per point-box test); brute force over boxes')
        del ins, cooc
        out['streaming_kernels'] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:          # reported at N=1 only
        cb, cls_cpu, pts_cpu, masks_cpu = cpu_baseline(args.cpu_sample, V, S, args.masks, flt)
        out['cpu_baseline'] = cb
        # the same sample through the HIP path must give the same labels
        got = f3d.default_context(local).project_vote_argmax(pts_cpu, views_np, masks_cpu, 133, 0.5, flt)
        out['parity_on_cpu_sample'] = bool(np.array_equal(got, cls_cpu))
    elif rank == 0:
        out['cpu_baseline'] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        for w_ in pending.values():
            w_.wait()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
