"""``split_into_instances`` of the reference's Fusion3DSeg/segUtils/cv.py (:402-500), host implementation.

This stage sits between voting and box merging (scope table row (f)#1, "next"): it is not a GPU kernel yet.  The
reference flood-fills with a Python list queue; here each flood fill is a level-synchronous BFS over a CSR copy of
the adjacency with NumPy frontiers -- same reachability, same processing order (classes in the given order, seed =
lowest remaining index), hence identical ids, info records and updated classes (pinned by tests/golden).
"""
import numpy as np


def _csr(adj, n):
    lens = np.fromiter((len(a) for a in adj), dtype=np.int64, count=n)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.concatenate([np.asarray(a, dtype=np.int64).reshape(-1) for a in adj]) if n and offs[-1] else np.zeros(0, np.int64)
    return offs, flat


def _reach(seed, seed_class, classes, offs, flat, visited):
    """Points of `seed_class` reachable from `seed` through same-class points (the reference's floodfill)."""
    visited[seed] = True
    frontier = np.array([seed], np.int64)
    parts = []
    while len(frontier):
        frontier = frontier[classes[frontier] == seed_class]
        if not len(frontier):
            break
        parts.append(frontier)
        starts, ends = offs[frontier], offs[frontier + 1]
        total = int((ends - starts).sum())
        if not total:
            break
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(ends - starts)[:-1]]), ends - starts) + np.arange(total)
        nb = np.unique(flat[idx])
        nb = nb[~visited[nb]]
        visited[nb] = True
        frontier = nb
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def split_into_instances(classes, adj, nclasses=133, instance_classes=None, minimum_points=1, verbose=False):
    """-> (instance ids [M], point ids [N], info list, updated classes [N]); see the reference docstring (:402-424)."""
    n = len(classes)
    classes = np.array(classes).copy()
    offs, flat = _csr(adj, n)
    allclasses = np.unique(classes)
    ids = np.zeros_like(classes)
    info, small_id = [], None
    if instance_classes is None:
        instance_classes, semantic_classes, ninst = allclasses, [], 0
        if (instance_classes == nclasses).any():
            instance_classes = instance_classes[instance_classes != nclasses]
            semantic_classes, ninst = [nclasses], 1
    else:
        instance_classes = np.array(instance_classes)
        semantic_classes = np.setdiff1d(allclasses, instance_classes)
        ninst = len(semantic_classes)
    for k in range(ninst if len(semantic_classes) else 0):
        c = semantic_classes[k]
        m = classes == c
        ids[m] = k
        if c == nclasses:
            small_id = k
        info.append({'id': k, 'isthing': False, 'category_id': int(c), 'area': int(m.sum())})
    for c in instance_classes:
        if verbose:
            print('splitting class:', c)
        remaining = classes == c
        order = np.nonzero(remaining)[0]
        cursor = 0
        while cursor < len(order):
            seed = order[cursor]
            if not remaining[seed]:
                cursor += 1
                continue
            cluster = _reach(seed, classes[seed], classes, offs, flat, np.zeros(n, bool))
            area = len(cluster)
            if area < minimum_points:
                cat = nclasses
                if small_id is None:
                    small_id = ninst
                    info.append({'id': ninst, 'isthing': True, 'category_id': int(cat), 'area': 0})
                    ninst += 1
                info[small_id]['area'] += area
                ids[cluster] = small_id
            else:
                cat = c
                info.append({'id': ninst, 'isthing': True, 'category_id': int(cat), 'area': int(area)})
                ids[cluster] = ninst
                ninst += 1
            remaining[cluster] = False
            classes[cluster] = cat
    return np.arange(ninst), ids, info, classes
