#!/usr/bin/env python3
"""Radius-graph adjacency (fusion.py:374-375) at scale: N synthetic cloud points (SURVEY 8(d) box), r = 2 * ds_radius.
Times the two GPU passes with HIP events on device-resident buffers, then feeds the graph to the same-class components
kernel, and times sklearn's KDTree.query_radius on a bounded sample of the same cloud for the CPU figure."""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'))
import torch  # noqa: E402
import f3d    # noqa: E402
from f3d import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--points', type=int, default=4_000_000)
ap.add_argument('--radius', type=float, default=0.05, help='ds_radius; the graph uses 2 * ds_radius as the reference does')
ap.add_argument('--cpu-sample', type=int, default=200_000)
args = ap.parse_args()
n, r = args.points, 2 * args.radius
P = synth.cloud(n)
ctx = f3d.default_context()
dev = torch.device('cuda', 0)
st = torch.cuda.Stream(dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(st):
    x = torch.from_numpy(P).to(dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nnz = C.c_int64(0)
    count = lambda: ctx._check(ctx._lib.f3d_radius_graph_count_dev(ctx._h, x.data_ptr(), f3d.F64, n, r, offs.data_ptr(), C.byref(nnz), st.cuda_stream))
    count(); st.synchronize()
    nb = torch.empty(nnz.value, dtype=torch.int32, device=dev)
    fill = lambda: ctx._check(ctx._lib.f3d_radius_graph_fill_dev(ctx._h, n, offs.data_ptr(), nb.data_ptr(), st.cuda_stream))
    fill(); st.synchronize()
    a, b, c = ev(), ev(), ev()
    a.record(st); count(); b.record(st); fill(); c.record(st); c.synchronize()
    t_count, t_fill = a.elapsed_time(b), b.elapsed_time(c)
    cls = torch.from_numpy(np.random.default_rng(0).choice([86, 114, 115, 133], n).astype(np.int64)).to(dev)
    par = torch.empty(n, dtype=torch.int32, device=dev)
    root = torch.empty(n, dtype=torch.int64, device=dev)
    a.record(st)
    ctx._check(ctx._lib.f3d_components_same_class_dev(ctx._h, cls.data_ptr(), n, offs.data_ptr(), nb.data_ptr(), par.data_ptr(), root.data_ptr(), st.cuda_stream))
    b.record(st); b.synchronize()
    t_cc = a.elapsed_time(b)
e = nnz.value
print(f'radius graph: {n} points, r = {r}: {e} directed edges ({e / n:.1f} per point); count {t_count:.1f} ms, fill {t_fill:.1f} ms '
      f'-> {n / (t_count + t_fill) / 1e3:.2f} M points/s, {e / (t_count + t_fill) / 1e6:.2f} G edges/s; same-class components on it {t_cc:.1f} ms')
m = min(args.cpu_sample, n)
from sklearn.neighbors import KDTree  # noqa: E402
t0 = time.perf_counter()
tree = KDTree(P)
t1 = time.perf_counter()
ref = tree.query_radius(P[:m], r=r)
t2 = time.perf_counter()
o = offs[:m + 1].cpu().numpy()
got = nb[:o[-1]].cpu().numpy()
ok = all(np.array_equal(np.sort(ref[i]), np.sort(got[o[i]:o[i + 1]])) for i in range(0, m, max(1, m // 2000)))
print(f'sklearn KDTree on the host: build {t1 - t0:.1f} s for {n} points, query_radius of the first {m} points {t2 - t1:.1f} s '
      f'-> {m / (t2 - t1) / 1e3:.1f} k points/s (1 core); rows compared on a stride: {"equal" if ok else "DIFFERENT"}')
