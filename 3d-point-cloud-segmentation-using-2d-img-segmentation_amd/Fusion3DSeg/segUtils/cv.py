"""``split_into_instances`` of the reference's Fusion3DSeg/segUtils/cv.py (:402-500) on the GPU.

The reference flood-fills, class by class, from the lowest remaining point index through same-class neighbours
(a Python list queue).  Here the flood fill is one GPU pass (f3d_components_same_class: lock-free union-find over the
CSR adjacency, every component rooted at its smallest index); the numbering that follows -- classes in the given
order, components by ascending seed, small clusters folded into one "unclassified" bucket created on first use -- is
reproduced with array operations, so ids, info records and updated classes are identical to the reference's
(pinned by tests/golden/split_instances.npz).

One stated difference: the adjacency is used as an undirected graph.  The reference follows neighbour lists as
directed edges; the two coincide for symmetric lists, which is what ``KDTree.query_radius`` (the only producer,
fusion.py:369-377) returns.
"""
import numpy as np

import f3d


def adjacency_to_csr(adj, n):
    """list / object array of neighbour index arrays -> (offsets int64 [n+1], neighbours int32 [E]); a CSR pair
    (as ``fusion.radius_adjacency(..., as_csr=True)`` returns it) passes through."""
    if isinstance(adj, tuple) and len(adj) == 2:
        offs, nbrs = np.ascontiguousarray(adj[0], np.int64), np.ascontiguousarray(adj[1], np.int32)
        if len(offs) != n + 1:
            raise ValueError('CSR adjacency: offsets must have n + 1 entries')
        return offs, nbrs
    lens = np.fromiter((len(a) for a in adj), dtype=np.int64, count=n)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    if n == 0 or offs[-1] == 0:
        return offs, np.zeros(0, np.int32)
    return offs, np.concatenate([np.asarray(a).reshape(-1) for a in adj]).astype(np.int32)


def split_into_instances(classes, adj, nclasses=133, instance_classes=None, minimum_points=1, verbose=False):
    """-> (instance ids [M], point ids [N], info list, updated classes [N]); see the reference docstring (:402-424)."""
    n = len(classes)
    classes = np.array(classes).copy()
    offs, nbrs = adjacency_to_csr(adj, n)
    ctx = f3d.default_context()
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

    root = ctx.components_same_class(classes, offs, nbrs) if n else np.zeros(0, np.int64)
    relabelled = False                                           # some cluster's class was rewritten since `root` was computed
    for c in instance_classes:
        if verbose:
            print('splitting class:', c)
        if relabelled and c == nclasses:                         # folded clusters now belong to this class: flood again
            root = ctx.components_same_class(classes, offs, nbrs)
            relabelled = False
        pts = np.nonzero(classes == c)[0]
        if not len(pts):
            continue
        seeds, inv, sizes = np.unique(root[pts], return_inverse=True, return_counts=True)      # ascending seed = visit order
        big = sizes >= minimum_points
        rank = np.cumsum(big) - 1                                # running index among the kept clusters
        comp_id = ninst + rank
        nbig = int(big.sum())
        if not big.all():
            first_small = int(np.argmax(~big))
            if small_id is None:                                 # the bucket takes the next id at its first use (:483-487)
                small_id = ninst + int(big[:first_small].sum())
                comp_id = comp_id + (np.arange(len(seeds)) > first_small)
                new_small = True
            else:
                new_small = False
            comp_id = np.where(big, comp_id, small_id)
        else:
            new_small = False
        # info records in id order: kept clusters, with the bucket's record spliced in where it was created
        recs = [{'id': int(i), 'isthing': True, 'category_id': int(c), 'area': int(a)} for i, a in zip(comp_id[big], sizes[big])]
        if new_small:
            pos = int(small_id - ninst)
            recs.insert(pos, {'id': int(small_id), 'isthing': True, 'category_id': int(nclasses), 'area': 0})
        info.extend(recs)
        if not big.all():
            info[small_id]['area'] += int(sizes[~big].sum())
            classes[pts[~big[inv]]] = nclasses                   # folded clusters become "unclassified" (:482,499)
            relabelled = True
        ids[pts] = comp_id[inv]
        ninst += nbig + (1 if new_small else 0)
    return np.arange(ninst), ids, info, classes
