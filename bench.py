#!/usr/bin/env python3
"""Headline benchmark: labelled points / second of the fused Fusion3DSeg hot path on MI355X.

One step = one pass of the hot path over one batch: every point of the cloud is projected into all
V views, sampled, voted and segmented (f3d_project_vote_argmax_dev), with the inputs already
resident in HBM.  Workload at N=1: BASELINE.json config C3 (10M points x 64 views, 1024x1024
masks).  With N>1 ranks (torchrun, one per GPU) the workload is C4 as BASELINE states it: the SAME
10M points, sharded by contiguous point ranges over the ranks (strong scaling); every rank owns
the masks of the V/N views it "produced" and each step all-gathers the masks over RCCL (the path's
one exchange step, double-buffered: the all-gather of step i+1 runs under the kernels of step i)
and then fuses its shard.  --weak gives every rank its own --points instead.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--filter] [--masks iid|block64] [--weak]

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
    ap.add_argument('--points', type=int, default=10_000_000, help='points of the cloud (C3/C4: 10M), sharded over the ranks; per rank with --weak')
    ap.add_argument('--weak', action='store_true', help='N > 1: every rank owns --points points of an N x larger cloud (weak scaling)')
    ap.add_argument('--exchange', default='pipelined', choices=['pipelined', 'chunked', 'coded'],
                    help='N > 1: pipelined = the all-gather of step i+1 overlaps the kernels of step i (double-buffered masks); chunked = inside '
                         'one step, the all-gather split into --chunks view chunks, chunk c+1 on the wire while chunk c votes; coded = chunked with the '
                         'mask coding sharded as well: every rank codes its own V/N masks, CODED planes are all-gathered')
    ap.add_argument('--chunks', type=int, default=2, help='view chunks per rank of --exchange chunked')
    ap.add_argument('--no-merge', action='store_true', help='skip the C5 bbox-merge leg (50M points, 4096 instances)')
    ap.add_argument('--views', type=int, default=64)
    ap.add_argument('--size', type=int, default=1024, help='mask width = height')
    ap.add_argument('--masks', default='block64', choices=['block64', 'block64x40', 'block64x96', 'block64x100', 'iid'])
    ap.add_argument('--filter', action='store_true', help='segment with the reference default filter_classes=[86,114,115]')
    ap.add_argument('--f32', action='store_true', help='store xyz as float32 (12 B/point) instead of float64')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the streaming-kernel measurements')
    ap.add_argument('--cpu-sample', type=int, default=5_000_000, help='points of the CPU-baseline slice (all views)')
    ap.add_argument('--sorted', action='store_true', help='experiment: hand the cloud over already in grid-cell order (host sort)')
    ap.add_argument('--sorted-morton', type=int, default=0, metavar='BITS', help='experiment: hand the cloud over in Morton order with BITS bits per axis (host sort), no sort in the step')
    ap.add_argument('--no-sort', action='store_true', help='do not cell-sort the cloud inside the step')
    ap.add_argument('--prepared', action='store_true', help='cell-sort once outside the timed loop and keep the sorted cloud resident')
    ap.add_argument('--overlap-sort', action='store_true', help='sort the cloud of step i+1 on a second stream while step i is fused (measured: no '
                    'gain, 1.43 vs 1.39 ms -- the fused kernel holds every wave slot of the chip, the sort kernels queue behind it)')
    return ap.parse_args()


THRESHOLD = float(os.environ.get('F3D_BENCH_THRESHOLD', '0.5'))      # (experiments only: the headline is BASELINE's 0.5)


def fuse_instance(labels_present, flt):
    """Which k_fuse instance the code book selects (csrc/f3d_fuse.hip: launch_fuse_t): codes = labels + 2 (no sample, rejected)."""
    codes = (3 + len(set(flt))) if (flt and len(flt) <= 8) else labels_present + 2
    if codes <= 12:
        return f'{codes} codes: dword bins (<= 12 codes)'
    if codes <= 48:
        return f'{codes} codes: packed 8-bit bins, view tables in LDS (<= 48 codes)'
    if codes <= 100:
        return f'{codes} codes: packed 8-bit bins, view tables in global memory (<= 100 codes)'
    return f'{codes} codes: packed 8-bit bins, any alphabet'


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
    out0 = np.empty(sample, np.int64)
    votes16 = np.empty((sample, 134), np.uint16)               # kept for the vote-row comparison (counts <= 64)
    dt = 0.0
    for s0 in range(0, sample, 250_000):
        t0 = time.perf_counter()
        votes = O.forward_votes(pts[s0:s0 + 250_000], K, q, t, masks, 10.0, ncols=134)
        out[s0:s0 + 250_000] = O.segment(votes, 133, 0.5, filter_classes)
        dt += time.perf_counter() - t0                         # the timed part = one pass of the path at the bench's setting
        out0[s0:s0 + 250_000] = O.segment(votes, 133, 0.0, filter_classes)      # threshold 0: every sampled point carries a real label
        votes16[s0:s0 + 250_000] = votes
    return dict(value=round(sample / dt, 1), unit='points/s', cores=1, kind='port',
                sample=f'{sample} points x {views_n} views in {dt:.1f} s, one NumPy process ({os.cpu_count()} host cpus visible, '
                       f'elementwise NumPy only -> 1 core); the path is linear in N'), out, pts, masks, (K, q, t), out0, votes16


def merge_leg(local):
    """C5's bbox-merge leg (BASELINE.json config 5, SURVEY 8(d) recipe): 50M points in 4096 Gaussian blobs (sigma 0.15 m, seed 3456),
    parent = id mod 8, through the drop-in merge_bb; beside it the oracle's literal restatement of the reference control flow
    (merge_intersecting_bb.py:103-137: O(B^2) refits and full-cloud scans) on a slice of the same recipe, one host core."""
    import copy
    import Fusion3DSeg.merge_intersecting_bb as M
    from oracle import np_ref as O

    def scene(B, n):
        rng = np.random.default_rng(3456)
        centres = rng.uniform([-5, -5, 0], [5, 5, 3], (B, 3))
        ids = rng.integers(1, B, n).astype(np.int64)
        pts = centres[ids] + rng.normal(size=(n, 3)) * 0.15
        return pts, ids, [{'id': k, 'category_id': 86, 'parent_id': k % 8, 'area': int((ids == k).sum())} for k in range(B)]

    B, n = 4096, 50_000_000
    pts, ids, info = scene(B, n)
    prof = {}
    keep = M._MergeState.__init__

    def spy(self, *a, **k):
        keep(self, *a, **k)
        prof['state'] = self
    M._MergeState.__init__ = spy
    try:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):                  # merge_bb prints its wall time like the reference; stdout carries the JSON line only
            info_h, ids_h = copy.deepcopy(info), ids.copy()            # merge_bb relabels its arguments in place, like the reference
            t0 = time.perf_counter()
            out_info, _ = M.merge_bb(None, info_h, ids_h, pts)
            dt = time.perf_counter() - t0
            del ids_h
    finally:
        M._MergeState.__init__ = keep
    st = prof['state'].prof
    # the same merge on a RESIDENT cloud (merge_bb_dev: device tensors in, nothing of the cloud or of the ids crosses PCIe)
    import torch
    dev = torch.device('cuda', local)
    dpts, dids = torch.from_numpy(pts).to(dev), torch.from_numpy(ids).to(dev)
    torch.cuda.synchronize(dev)
    M._MergeState.__init__ = spy
    try:
        with contextlib.redirect_stdout(sys.stderr):
            info_d = copy.deepcopy(info)
            t0 = time.perf_counter()
            dev_info, dev_ids = M.merge_bb_dev(info_d, dids, dpts)
            torch.cuda.synchronize(dev)
            dt_dev = time.perf_counter() - t0
    finally:
        M._MergeState.__init__ = keep
    st_dev = prof['state'].prof
    dev_same = [(d['id'], d['area']) for d in dev_info] == [(d['id'], d['area']) for d in out_info]
    del pts, ids, dpts, dids, dev_ids
    Bs, ns = 384, 192_000
    p2, i2, f2 = scene(Bs, ns)
    t0 = time.perf_counter()
    o_info, o_ids = O.merge_bb(copy.deepcopy(f2), i2.copy(), p2)
    dtc = time.perf_counter() - t0
    with contextlib.redirect_stdout(sys.stderr):
        g_info, g_ids = M.merge_bb(None, copy.deepcopy(f2), i2.copy(), p2)                  # the product's own (GPU) fit against the oracle's
    same = bool(np.array_equal(g_ids, o_ids) and [(d['id'], d['area']) for d in g_info] == [(d['id'], d['area']) for d in o_info])
    return dict(workload=f'C5 merge: {n} points, {B} instances (Gaussian blobs, parent = id mod 8) -> {len(out_info)} entries',
                seconds=round(dt, 3), points_per_s=round(n / dt, 1),
                breakdown_s={k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()},
                resident=dict(seconds=round(dt_dev, 3), points_per_s=round(n / dt_dev, 1), same_entries_as_host_call=bool(dev_same),
                              breakdown_s={k: (round(v, 3) if isinstance(v, float) else v) for k, v in st_dev.items()},
                              note='merge_bb_dev: cloud and ids are device tensors before the clock starts'),
                algorithmic_bytes_per_scan=24 * n,
                note='host-pointer call: the 1.2 GB cloud is uploaded inside the timed region (breakdown_s.upload); prefilter = extremes, inner hulls, '
                     'strict-inside filter and ordered compaction, all on the device (f3d_obb_candidates_dev); fit = f3d_obb_fit_dev for every instance in one '
                     'launch (hull vertices with certified signs, PCA, Jacobi) + refits after merges (nfit_deferred = fits the kernel handed to the host); '
                     'scans = k_points_in_obb launches',
                cpu_baseline=dict(value=round(dtc, 2), unit='s', cores=1, kind='port',
                                  sample=f'oracle/np_ref.py::merge_bb (reference control flow, O(B^2) refits and scans) on {ns} points, {Bs} '
                                         f'instances of the same recipe; its cost grows like B^2 N / 8, i.e. ~{dtc * (B / Bs) ** 2 * (n / ns) / 3600:.0f} h '
                                         f'extrapolated to the full C5 shape',
                                  same_result_as_hip_on_sample=same))


def end_to_end_leg(torch, ctx, dev, stream, xyz, dtype, n, views, views_np, V, S, flt, flags, perm_ptr, steps=3):
    """BASELINE config 3 as written: "end-to-end incl. the 2D backbone on PyTorch-ROCm".  Per step: V frames through a
    PyTorch-ROCm network (a STAND-IN with OneFormer's contract, f3d/standin.py -- OneFormer itself is absent), the network's
    logits -> mask kernel writing plane j of ONE uint8 [V,H,W] device tensor (f3d_sem_logits_to_masks_dev; no PNG, no D2H, no
    synchronisation), and the fused call reading that tensor.  torch's sync debug mode is armed ('error') over the steps: any
    synchronising torch call (.cpu(), .item(), ...) inside would raise."""
    import get2DSeg
    from f3d.standin import StandInSegNet, synthetic_frames
    from oracle import np_ref as O
    from f3d import synth
    B = 8
    net = StandInSegNet().to(dev).eval()
    frames = synthetic_frames(V, S, S, dev)
    masks = torch.empty((V, S, S), dtype=torch.uint8, device=dev)
    classes = torch.empty(n, dtype=torch.int64, device=dev)
    nb = (V + B - 1) // B
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nb)]
    ev_f = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def step(timed=False):
        for k, j in enumerate(range(0, V, B)):
            if timed:
                ev[k][0].record(stream)
            sem = net(frames[j:j + B])
            if timed:
                ev[k][1].record(stream)
            get2DSeg.sem_to_mask_device(sem, masks[j:j + B], 0.017)
            if timed:
                ev[k][2].record(stream)
        if timed:
            ev_f[0].record(stream)
        ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, masks.data_ptr(), S, S, 133, 0.5, flt,
                                    classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
        if timed:
            ev_f[1].record(stream)

    step(); step()                                            # warm-up (GEMM selection, allocator)
    torch.cuda.synchronize()

    def arm(mode):
        try:
            torch.cuda.set_sync_debug_mode(mode)
            return True
        except Exception:                                     # not supported by this torch build: the property is then unchecked
            return False
    armed = arm('error')
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        arm('default')
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        arm('error')
        step(timed=True)
    finally:
        arm('default')
    torch.cuda.synchronize()
    t_net = sum(e[0].elapsed_time(e[1]) for e in ev)
    t_mask = sum(e[1].elapsed_time(e[2]) for e in ev)
    t_fuse = ev_f[0].elapsed_time(ev_f[1])
    ctx.take_device_error(stream.cuda_stream)
    # parity: (1) the masks against the reference's own statements (get2DSeg.py:110-118) in torch on the same logits, outside
    # the 1e-5 relative band around the threshold that sem_mask.npz's rule states; (2) labels of a sample against the oracle
    # fed the masks the device produced
    sem = net(frames[:2])
    want = sem.argmax(dim=1)
    pmax = torch.amax(torch.nn.Softmax(dim=1)(sem), dim=1)
    want[pmax < 0.017] = 133
    clear = (pmax.double() - 0.017).abs() > 1e-5 * 0.017
    got = torch.empty((2, S, S), dtype=torch.uint8, device=dev)
    get2DSeg.sem_to_mask_device(sem, got, 0.017)
    masks_ok = bool(torch.equal(got[clear].long(), want[clear]) and float(clear.float().mean()) > 0.9999)
    masks_np = masks.cpu().numpy()
    K = np.array([[800., 0, S / 2], [0, 800., S / 2], [0, 0, 1]])
    q, t = synth.ring_views(V)
    sub = np.random.default_rng(11).choice(n, min(n, 100_000), replace=False)
    pts = xyz[torch.from_numpy(sub).to(dev)].double().cpu().numpy()
    votes = O.forward_votes(pts, K, q, t, masks_np, 10.0, ncols=134)
    want_cls, want_cls0 = O.segment(votes, 133, 0.5, flt), O.segment(votes, 133, 0.0, flt)
    got_cls = classes.cpu().numpy()[sub]
    ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, masks.data_ptr(), S, S, 133, 0.0, flt,
                                classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)   # threshold 0: real labels
    stream.synchronize()
    got_cls0 = classes.cpu().numpy()[sub]
    return dict(workload=f'C3 end-to-end: {V} frames {S}x{S} -> stand-in 2D network (PyTorch-ROCm, bf16 GEMMs; NOT OneFormer, which is absent: '
                         f'f3d/standin.py) -> logits [133,{S},{S}] f32 -> f3d_sem_logits_to_masks_dev -> uint8 [V,H,W] device tensor -> '
                         f'f3d_project_vote_argmax_dev over {n} points',
                ms_per_step=round(dt * 1e3, 3), points_per_s=round(n / dt, 1),
                stage_ms=dict(network_standin=round(t_net, 3), logits_to_mask=round(t_mask, 3), fused_3d=round(t_fuse, 3)),
                frames_per_batch=B, png_round_trip=False,
                no_host_sync=('torch.cuda.set_sync_debug_mode("error") armed over the steps: no synchronising torch call, no D2H; the library calls are _dev entry points (enqueue only)'
                              if armed else 'unchecked: torch.cuda.set_sync_debug_mode unavailable'),
                masks_equal_reference_statements=masks_ok, mask_labels_present=int(len(np.unique(masks_np))),
                low_confidence_pixel_fraction=round(float((masks_np == 133).mean()), 4),
                labels_equal_oracle_on_sample=bool(np.array_equal(got_cls, want_cls) and np.array_equal(got_cls0, want_cls0)), sample_points=len(sub),
                real_label_fraction=dict(threshold_0p5=round(float((want_cls != 133).mean()), 4), threshold_0p0=round(float((want_cls0 != 133).mean()), 4)))


def variants_leg(torch, ctx, dev, stream, xyz, dtype, n, views, V, S, flags, perm_ptr, steps=5):
    """The other SURVEY 8(d) workloads through the same timed step (caller-order cloud, in-step sort): mask variants iid (134 labels
    in every tile: the any-alphabet instance of k_fuse) and a 40-label blocky alphabet (packed 8-bit bins), and the reference's
    default filter list; ms per step each."""
    from f3d import synth
    out = {}
    classes = torch.empty(n, dtype=torch.int64, device=dev)
    for name, kind, flt in [('iid', 'iid', None), ('block64x40', 'block64x40', None), ('block64x100', 'block64x100', None),
                            ('block64_filter_86_114_115', 'block64', [86, 114, 115])]:
        m = torch.from_numpy(synth.masks(V, S, S, kind)).to(dev)

        def fn():
            ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, m.data_ptr(), S, S, 133, 0.5, flt,
                                        classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
        tk = time_kernel(torch, fn, steps, stream)
        ctx.take_device_error(stream.cuda_stream)
        out[name] = dict(ms_per_step=round(tk * 1e3, 4), points_per_s=round(n / tk, 1), labels_in_masks=int(len(np.unique(m[:4].cpu().numpy()))),
                         filter_classes=flt)
        del m
    # threshold 0.0 on the headline's masks: ~90 % of the points carry a real label instead of 0.1 % -- the label vector is pre-filled with
    # "unlabelled" and only real labels are scattered back to caller order, so this is the expensive end of that store
    m = torch.from_numpy(synth.masks(V, S, S, 'block64')).to(dev)

    def fn0():
        ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, views.data_ptr(), V, m.data_ptr(), S, S, 133, 0.0, None,
                                    classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
    tk = time_kernel(torch, fn0, steps, stream)
    ctx.take_device_error(stream.cuda_stream)
    out['block64_threshold_0'] = dict(ms_per_step=round(tk * 1e3, 4), points_per_s=round(n / tk, 1), labels_in_masks=8, filter_classes=None,
                                      real_label_fraction=round(float((classes != 133).float().mean().item()), 4))
    del m
    # the C5 shape of the fused call on this cloud: 256 views (four 64-view groups: the instances that read their view tables from global
    # memory; deferred points carry one mask of open views per group)
    import f3d
    V5 = 256
    K = np.array([[800., 0, S / 2], [0, 800., S / 2], [0, 0, 1]])
    q5, t5 = synth.ring_views(V5)
    v5 = torch.from_numpy(f3d.views_build(K, S, S, q5, t5, 10.0)).to(dev)
    m5 = torch.from_numpy(synth.masks(V5, S, S, 'block64')).to(dev)

    def fn5():
        ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, n, v5.data_ptr(), V5, m5.data_ptr(), S, S, 133, 0.5, None,
                                    classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
    tk = time_kernel(torch, fn5, 3, stream)
    ctx.take_device_error(stream.cuda_stream)
    d5 = ctx.fuse_deferred(stream.cuda_stream)
    out['block64_256_views'] = dict(ms_per_step=round(tk * 1e3, 4), points_per_s=round(n / tk, 1), point_views_per_s=round(n * V5 / tk, 1),
                                    labels_in_masks=int(len(np.unique(m5[:4].cpu().numpy()))), filter_classes=None,
                                    deferred_points=dict(to_float64_tier=d5[0], to_exact_arithmetic=d5[1]))
    del m5, v5
    # what strong scaling can reach: the step of one rank's share of this cloud -- its first n/2, n/4, n/8 points against all
    # the views -- without any exchange; the fixed work per launch and per call does not shrink with the share
    m = torch.from_numpy(synth.masks(V, S, S, 'block64')).to(dev)
    shares = {}
    for parts in (2, 4, 8):
        ns = n // parts

        def fns():
            ctx.project_vote_argmax_dev(xyz.data_ptr(), dtype, ns, views.data_ptr(), V, m.data_ptr(), S, S, 133, 0.5, None,
                                        classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=None if perm_ptr is None else perm_ptr)
        if perm_ptr is None:                                                   # (a prepared layout's permutation covers the whole cloud only)
            tk = time_kernel(torch, fns, 10, stream)
            shares[f'1/{parts}'] = dict(points=ns, ms_per_step=round(tk * 1e3, 4))
    ctx.take_device_error(stream.cuda_stream)
    if shares:
        out['share_of_the_cloud_no_exchange'] = shares
    del m
    return out


def c1_leg(local):
    """C1 (100k points x 4 views, 512x512) is CPU-timed in full -- the oracle, one process -- next to the
    same inputs through the HIP path (host-pointer call: upload and download inside its time)."""
    import f3d
    from f3d import synth
    from oracle import np_ref as O
    sc = synth.scene('C1')
    t0 = time.perf_counter()
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, 0.5, [86, 114, 115])
    dtc = time.perf_counter() - t0
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    hctx = f3d.default_context(local)
    hctx.project_vote_argmax(sc['points'][:1000], views, sc['masks'], 133, 0.5, [86, 114, 115])
    t0 = time.perf_counter()
    got = hctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, 0.5, [86, 114, 115])
    dtg = time.perf_counter() - t0
    return dict(value=round(len(want) / dtc, 1), unit='points/s', cores=1, kind='port', seconds=round(dtc, 4),
                sample='C1 in full: 100000 points x 4 views, 512x512 block-64 masks, threshold 0.5, filter [86,114,115]; oracle/np_ref.py, one process',
                hip_host_pointer_call_s=round(dtg, 4), labels_equal=bool(np.array_equal(got, want)))


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

    V, S = args.views, args.size
    if world > 1 and not args.weak:
        lo, hi = sharding.point_bounds(args.points, rank, world)          # C4: one cloud, contiguous point ranges
        n, total_points, scaling = hi - lo, args.points, 'strong'
    else:
        lo, n, total_points, scaling = 0, args.points, args.points * world, ('weak' if args.weak and world > 1 else 'strong')
    if n <= 0:
        raise SystemExit('more ranks than points')
    flt = [86, 114, 115] if args.filter else None
    K = np.array([[800., 0, S / 2], [0, 800., S / 2], [0, 0, 1]])
    q, t = synth.ring_views(V)
    views_np = f3d.views_build(K, S, S, q, t, 10.0)
    if scaling == 'strong' and world > 1:
        xyz_np = synth.cloud(args.points, dtype=np.float32 if args.f32 else np.float64)[lo:lo + n]      # this rank's range of THE cloud
    else:
        xyz_np = synth.cloud(n, dtype=np.float32 if args.f32 else np.float64, shard=rank)
    if args.sorted:
        cell = np.floor((xyz_np.astype(np.float64) - np.array([-5, -5, 0])) / 0.25).astype(np.int64)
        xyz_np = xyz_np[np.argsort((cell[:, 0] * 64 + cell[:, 1]) * 16 + cell[:, 2], kind='stable')]
    if args.sorted_morton < 0:               # experiment: the device's 16-bit cell order (6, 6, 4 bits), then every 128-point wave-tile ordered by a fine Morton key
        def morton(q, bits):
            key = np.zeros(len(q), np.int64)
            for level in range(max(bits) - 1, -1, -1):
                for c in range(3):
                    if bits[c] > level:
                        key = (key << 1) | ((q[:, c] >> level) & 1)
            return key
        rel = (xyz_np.astype(np.float64) - np.array([-5, -5, 0])) / np.array([10.0, 10.0, 3.0])
        coarse = morton(np.clip((rel * np.array([64, 64, 16])).astype(np.int64), 0, [63, 63, 15]), (6, 6, 4))
        fine = morton(np.clip((rel * 1024).astype(np.int64), 0, 1023), (10, 10, 10))
        order = np.argsort(coarse, kind='stable')
        grp = np.arange(len(order)) // (-args.sorted_morton)
        order = order[np.lexsort((fine[order], grp))]
        xyz_np = xyz_np[order]
        args.sorted = True
    if args.sorted_morton > 0:
        b = args.sorted_morton
        q = np.clip(((xyz_np.astype(np.float64) - np.array([-5, -5, 0])) / 10.0 * (1 << b)).astype(np.int64), 0, (1 << b) - 1)
        key = np.zeros(len(q), np.int64)
        for level in range(b - 1, -1, -1):
            for c in range(3):
                key = (key << 1) | ((q[:, c] >> level) & 1)
        xyz_np = xyz_np[np.argsort(key, kind='stable')]
        args.sorted = True
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
                                        133, THRESHOLD, flt, classes.data_ptr(), None, stream.cuda_stream, flags=flags, perm_ptr=perm_ptr)
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

    chunked = use_dist and args.exchange in ('chunked', 'coded')
    coded_x = use_dist and args.exchange == 'coded'
    if chunked:
        if overlap or args.prepared:
            raise SystemExit('--exchange chunked takes the caller-order cloud (the sort is part of the first chunk)')
        vc, order = sharding.chunk_layout(V, world, args.chunks)
        views_chunked = torch.from_numpy(np.ascontiguousarray(views_np[order])).to(dev)
        engine = sharding.HipChunkEngine(ctx, xyz, dtype, n, views_chunked, S, S, 133, 0.5, flt, classes, flags=flags)
        views = views_chunked                    # the gather buffer holds the planes in this order: one-shot calls on it use the same
        if coded_x:
            coded_buf = torch.empty((V, engine.coded_plane_bytes()), dtype=torch.uint8, device=dev)

    def step():
        if not use_dist:
            fuse()
            return
        if coded_x:                              # coded planes travel; no rank holds another rank's raw masks
            sharding.overlapped_labels_coded(dist, engine, masks_shard, coded_buf, args.chunks)
            return
        if chunked:                              # masks_full is the gather buffer, planes in chunk order
            sharding.overlapped_labels(dist, engine, masks_shard, masks_full, args.chunks)
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
    chunk_check = None
    if chunked:                                             # the chunked step against one call over all views, same inputs
        stepped = classes.clone()
        if coded_x:                                         # (the raw planes of the other ranks, for this check only, in the buffer's order)
            for c in range(args.chunks):
                vcn = V // world // args.chunks
                part = masks_full.view(args.chunks, world, vcn, S, S)[c]
                dist.all_gather_into_tensor(part.reshape(-1), masks_shard[c * vcn:(c + 1) * vcn].contiguous().view(-1))
        fuse(masks_full)
        stream.synchronize()
        chunk_check = bool(torch.equal(stepped, classes))

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
    # HBM traffic of the dominant kernel from the PMC passes of THIS build (scripts/pmc_custom.sh; separate rocprofv3 --pmc runs):
    # the CSV names the library it was taken with, a different build gets null
    traffic, traffic_src = None, 'no PMC pass of this build under profiles/ (the CSV names another libf3d_hip.so)'
    import hashlib
    lib_sha = hashlib.sha256(f3d.library_path().read_bytes()).hexdigest()[:16]
    pmc = ROOT / 'profiles' / 'r03_pmc_by_kernel.csv'
    if (pmc.is_file() and n == 10_000_000 and V == 64 and S == 1024 and not args.f32 and world == 1 and args.masks == 'block64' and
            not (args.prepared or args.sorted or args.no_sort or args.filter)):      # the counters were taken on the default workload only
        lines = pmc.read_text().splitlines()
        if lines and lines[0].strip() == f'# lib_sha16={lib_sha}':
            vals = {}
            for line in lines[1:]:
                parts = line.split(',')
                if len(parts) >= 3 and parts[0] == 'k_fuse':
                    vals[parts[1]] = float(parts[2])
            if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
                traffic = int((vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)
                traffic_src = (f'profiles/r03_pmc_by_kernel.csv (lib {lib_sha}): (FETCH_SIZE + WRITE_SIZE) KB per k_fuse launch, raw (the x2 '
                               'FETCH_SIZE correction is calibrated for wide streams, not for 1-byte gathers; L2 misses incl. Infinity-Cache hits)')
    roofline = dict(bound='hbm', achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                    kernel='k_fuse', kernel_ms=round(t_kernel * 1e3, 4), sort_ms=round(t_sort * 1e3, 4),
                    algorithmic_bytes=abytes,
                    reference_flop_frac=round(FLOP_PER_POINT_VIEW * n * V / t_kernel / (FP64_VALU_PEAK_TFLOPS * 1e12), 4),
                    note='kernel_ms = the whole fused call (label presence + mask coding + k_fuse + float64 middle tier + exact '
                         'kernel), HIP events on its stream.  The fused V-view path is instruction-issue / gather bound, not HBM '
                         'bound (SURVEY 8(d): HBM-fraction ceiling ~7% at V=64).  reference_flop_frac = 82 flop of the REFERENCE '
                         'formulation x N x V / t / 78.6 TF fp64 -- NOT a utilisation: ~55% of the (wave, view) pairs are culled and '
                         'never executed, and the executed arithmetic is float32 (profiles/ has the executed-instruction counters)')

    out = None
    if rank == 0:
        out = dict(metric='labelled points/sec (10M pts x 64 views) at 1/2/4/8 GPU; % HBM roofline',
                   value=round(total_points * args.steps / elapsed, 1), unit='points/s', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
                   higher_is_better=True, scaling=scaling, vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(workload=(f'C3: {total_points} points x {V} ring views' if world == 1 else
                                         f'C4: {total_points} points sharded over {world} GPUs ({scaling} scaling) x {V} ring views') +
                                        f', {S}x{S} {args.masks} uint8 masks, nclasses=133, threshold=0.5, filter_classes={flt}; '
                                        f'fused project->sample->vote->segment',
                               points_total=total_points, points_per_gpu=n, views=V, mask_hw=[S, S], xyz_storage='f32' if args.f32 else 'f64', cloud_layout=layout,
                               chunked_step_equals_one_call=chunk_check,
                               deferred_points=dict(to_float64_tier=deferred[0], to_exact_kernel=deferred[1]),
                               mask_alphabet=dict(labels_present=int(len(np.unique(masks_np[:8]))),
                                                  k_fuse_instance=fuse_instance(int(len(np.unique(masks_np[:8]))), flt),
                                                  note='block64 = the 8-label alphabet of SURVEY 8(d): the small-alphabet (dword-bin) instance, the fastest '
                                                       'of the four; `variants` has iid / 40 / 100 labels and the filter list through the same step'),
                               exchange='none' if not use_dist else (f'RCCL all_gather of {V // world} masks/rank per step, double-buffered and overlapped with the previous step' if not chunked else
                                                                     f'RCCL all_reduce(MAX) of the label-presence bytes + {args.chunks} all_gathers of {V // world // args.chunks} {"CODED " if coded_x else ""}masks/rank per step; chunk c+1 on the wire while chunk c votes (vote state carried in HBM)' + (' -- each rank codes only its own masks' if coded_x else ''))),
                   roofline=roofline)

    # secondary, HBM-streaming kernels of the same path (not part of `value`)
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        # the per-point streaming kernels run on a 50M-point cloud (1.2 GB of xyz, 1.65 GB per pass): far beyond the 256 MiB
        # Infinity Cache, so that GB/s is HBM traffic and not cache hits across the timed repeats
        nbig = 50_000_000
        big = torch.from_numpy(synth.cloud(nbig, dtype=np.float32 if args.f32 else np.float64)).to(dev)
        uv = torch.empty((2, nbig), dtype=torch.int32, device=dev)
        ins = torch.empty(nbig, dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.project_view_dev(big.data_ptr(), dtype, nbig, views_np[0], uv.data_ptr(), ins.data_ptr(),
                                                             stream.cuda_stream), 5, stream)
        b = (xyz_b + 8 + 1) * nbig
        extras['project_view (a2+a4, 1 view)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                        bytes_per_point=xyz_b + 9, points=nbig)
        del uv
        ns = min(n, 4_000_000)
        votes = torch.zeros((ns, 134), dtype=torch.float64, device=dev)
        votes.view(-1)[::7] = 3.0
        cls2 = torch.empty(ns, dtype=torch.int64, device=dev)
        tk = time_kernel(torch, lambda: ctx.segment_votes_dev(votes.data_ptr(), ns, 134, 133, 0.5, None, cls2.data_ptr(), stream.cuda_stream), 10, stream)
        b = (134 * 8 + 8) * ns
        extras['segment_votes (a8)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                             bytes_per_point=1080, points=ns)
        del votes, cls2
        # a7: the uv2pt scatter vote, 1024x1024 lookups, ~70 % valid, 4M points: frame by frame (3 launches + a 16 MiB memset each)
        # against the batched call (one launch pair for all frames), on lookups without any duplication (random: every pixel its own
        # point -- the worst case, two scattered atomics per pixel whatever the schedule) and on patch-structured ones (every fused
        # point owns a 5x5 pixel patch, as Fusion.fuse writes them: duplicates die in LDS)
        hw = S * S
        rng = np.random.default_rng(7)
        Fv = 16
        votes = torch.zeros((ns, 134), dtype=torch.float64, device=dev)
        mv = masks_full[:Fv].reshape(Fv, -1)
        for kind in ('random', 'patch5x5'):
            if kind == 'random':
                lut_np = rng.integers(0, ns, (Fv, hw)).astype(np.int32)
            else:
                cells = rng.integers(0, ns, (Fv, (S + 4) // 5, (S + 4) // 5)).astype(np.int32)
                lut_np = np.repeat(np.repeat(cells, 5, 1), 5, 2)[:, :S, :S].reshape(Fv, hw).copy()
            lut_np[rng.random((Fv, hw)) < 0.3] = -1
            lut = torch.from_numpy(lut_np).to(dev)
            nvalid = int((lut_np != -1).sum()) // Fv

            def per_frame():
                for f in range(Fv):
                    ctx.vote_uv2pt_dev(lut[f].data_ptr(), mv[f].data_ptr(), hw, votes.data_ptr(), ns, 134, stream.cuda_stream)
            tk1 = time_kernel(torch, per_frame, 3, stream) / Fv
            tkb = time_kernel(torch, lambda: ctx.vote_uv2pt_batch_dev(lut.data_ptr(), mv.data_ptr(), Fv, S, S, votes.data_ptr(), ns, 134, stream.cuda_stream),
                              3, stream) / Fv
            b = 5 * hw + 16 * nvalid
            extras[f'vote_uv2pt (a7, 1024x1024 frames, {kind} lookups)'] = dict(
                per_frame_ms=round(tk1 * 1e3, 4), batched_ms_per_frame=round(tkb * 1e3, 4), frames_per_s_per_frame_api=round(1 / tk1, 1),
                frames_per_s_batched=round(1 / tkb, 1), speedup=round(tk1 / tkb, 2), GBps_batched=round(b / tkb / 1e9, 1),
                hbm_frac_batched=round(b / tkb / 1e9 / HBM_PEAK_GBS, 4),
                note='algorithmic bytes: 5 B/pixel streamed + 8 B read + 8 B write per valid lookup (scatter: line-granular traffic is ~8x that)')
            del lut
        del votes
        # a9: logits -> mask for one 133 x 1024 x 1024 image
        sem = torch.randn((133, S, S), dtype=torch.float32, device=dev)
        mk = torch.empty((S, S), dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.sem_logits_to_mask_dev(sem.data_ptr(), 133, hw, 0.017, 133, mk.data_ptr(), stream.cuda_stream), 10, stream)
        b = (133 * 4 + 1) * hw
        extras['sem_logits_to_mask (a9, 133x1024x1024)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                                  bytes_per_pixel=533)
        del sem, mk
        # a4 single-purpose streaming kernel, same 50M-point working set
        F = f3d.view_fields(views_np)
        pp, pn = np.ascontiguousarray(F['plane_pt'][0]), np.ascontiguousarray(F['plane_n'][0])
        tk = time_kernel(torch, lambda: ctx._check(ctx._lib.f3d_inside_polyhedra_dev(ctx._h, big.data_ptr(), dtype, nbig, pp.ctypes.data, pn.ctypes.data, 5,
                                                                                   ins.data_ptr(), stream.cuda_stream)), 5, stream)
        b = (xyz_b + 1) * nbig
        extras['inside_polyhedra (a4, 5 planes)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                         bytes_per_point=xyz_b + 1, points=nbig)
        if not args.f32:                                    # a1: the quaternion sandwich alone, same working set (24 B in + 24 B out)
            rot = torch.empty_like(big)
            tk = time_kernel(torch, lambda: ctx.rotate_dev(big.data_ptr(), nbig, views_np[0][49:53], rot.data_ptr(), stream.cuda_stream), 5, stream)
            extras['rotate (a1)'] = dict(ms=round(tk * 1e3, 4), GBps=round(48 * nbig / tk / 1e9, 1), hbm_frac=round(48 * nbig / tk / 1e9 / HBM_PEAK_GBS, 4),
                                         bytes_per_point=48, points=nbig)
            del rot
        del big, ins
        ins = torch.empty(n, dtype=torch.uint8, device=dev)
        # (f)#3: one 1024x1024 16-bit depth frame -> world points
        dep = torch.randint(0, 6000, (S, S), device=dev, dtype=torch.int16)       # < 32768: the same bits as uint16
        wpts = torch.empty((S * S, 3), dtype=torch.float64, device=dev)
        Kd = np.ascontiguousarray(K, dtype=np.float64); qd = np.array([0.5, 0.5, -0.5, 0.5]); td = np.array([0.25, -0.5, 1.0])
        tk = time_kernel(torch, lambda: ctx._check(ctx._lib.f3d_unproject_depth_dev(ctx._h, dep.data_ptr(), 2, S, S, Kd.ctypes.data, 1000.0, qd.ctypes.data,
                                                                                    td.ctypes.data, wpts.data_ptr(), stream.cuda_stream)), 10, stream)
        b = 26 * S * S
        extras['unproject_depth ((f)#3, 1024x1024 u16 frame)'] = dict(ms=round(tk * 1e3, 4), GBps=round(b / tk / 1e9, 1), hbm_frac=round(b / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                                       bytes_per_pixel=26)
        Fd = 16                                              # the same as ONE launch over 16 frames
        depb = torch.randint(0, 6000, (Fd, S, S), device=dev, dtype=torch.int16)
        wptb = torch.empty((Fd, S * S, 3), dtype=torch.float64, device=dev)
        qb, tb = np.tile(qd, (Fd, 1)), np.tile(td, (Fd, 1))
        tkb = time_kernel(torch, lambda: ctx.unproject_depth_batch_dev(depb.data_ptr(), 2, Fd, S, S, Kd, qb, tb, wptb.data_ptr(), 1000.0, stream.cuda_stream), 5, stream) / Fd
        extras['unproject_depth ((f)#3, 1024x1024 u16 frame)'].update(batched_ms_per_frame=round(tkb * 1e3, 4), GBps_batched=round(b / tkb / 1e9, 1),
                                                                     hbm_frac_batched=round(b / tkb / 1e9 / HBM_PEAK_GBS, 4), frames_per_batch=Fd)
        del dep, wpts, depb, wptb
        # a10/a11: all points x 64 oriented boxes, membership co-occurrence only
        boxes = np.zeros((64, 15)); boxes[:, 0:3] = rng.uniform([-5, -5, 0], [5, 5, 3], (64, 3))
        boxes[:, 3:12] = np.eye(3).reshape(-1); boxes[:, 12:15] = 0.8
        cooc = torch.zeros((64, 64), dtype=torch.uint8, device=dev)
        tk = time_kernel(torch, lambda: ctx.points_in_obb_dev(xyz.data_ptr(), dtype, n, boxes, None, cooc.data_ptr(), stream.cuda_stream), 5, stream)
        extras['points_in_obb (a10/a11, 64 boxes)'] = dict(ms=round(tk * 1e3, 4), point_box_tests_per_s=round(64 * n / tk, 1),
                                                           GBps=round(xyz_b * n / tk / 1e9, 1), hbm_frac=round(xyz_b * n / tk / 1e9 / HBM_PEAK_GBS, 4),
                                                           note='8..64 boxes: a 2048-cell table over the boxes\' bounds (k_obb_cells), a point runs the float64 in-box test (27 flop) only for the boxes of its cell; other box counts: per (wave, box) a float32 bounds test first')
        del ins, cooc
        out['streaming_kernels'] = extras

    if rank == 0 and world == 1 and not args.no_extras and not (args.prepared or args.sorted or overlap):
        out['variants'] = variants_leg(torch, ctx, dev, stream, xyz, dtype, n, views, V, S, flags, perm_ptr)
        if 'share_of_the_cloud_no_exchange' in out['variants']:
            out['share_of_the_cloud_no_exchange'] = out['variants'].pop('share_of_the_cloud_no_exchange')
        out['c3_end_to_end'] = end_to_end_leg(torch, ctx, dev, stream, xyz, dtype, n, views, views_np, V, S, flt, flags, perm_ptr)

    if rank == 0 and world == 1 and not args.no_merge and not args.no_extras:
        out['merge_bb'] = merge_leg(local)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:          # reported at N=1 only
        from oracle import np_ref as O
        cb, cls_cpu, pts_cpu, masks_cpu, (K, q, t), want0, wvotes = cpu_baseline(args.cpu_sample, V, S, args.masks, flt)
        out['cpu_baseline'] = cb
        out['cpu_baseline_c1'] = c1_leg(local)
        # the same sample through the HIP path must give the same labels -- at the bench's threshold AND at threshold 0, where
        # every sampled point carries a real label (at 0.5 with independent random masks ~99.9 % are 133 = "unclassified") -- and
        # the same vote rows; the WHOLE CPU slice at both settings
        hctx = f3d.default_context(local)
        got = hctx.project_vote_argmax(pts_cpu, views_np, masks_cpu, 133, 0.5, flt)
        got0, gvotes = hctx.project_vote_argmax(pts_cpu, views_np, masks_cpu, 133, 0.0, flt, return_votes=True)
        votes_equal = bool(np.array_equal(gvotes, wvotes))
        out['parity_on_cpu_sample'] = bool(np.array_equal(got, cls_cpu) and np.array_equal(got0, want0) and votes_equal)
        out['parity_detail'] = dict(threshold_0p5=dict(points=len(pts_cpu), equal=bool(np.array_equal(got, cls_cpu)),
                                                       real_label_fraction=round(float((cls_cpu != 133).mean()), 5)),
                                    threshold_0p0=dict(points=len(pts_cpu), equal=bool(np.array_equal(got0, want0)), votes_equal=votes_equal,
                                                       real_label_fraction=round(float((want0 != 133).mean()), 5)))
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
