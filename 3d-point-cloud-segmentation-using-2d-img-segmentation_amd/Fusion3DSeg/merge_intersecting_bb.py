"""Drop-in for the reference's Fusion3DSeg/merge_intersecting_bb.py::merge_bb.

The reference decides that two instances overlap when their oriented boxes share at least one cloud point and
does so with O(B^2) full-cloud scans plus Python list intersections (reference :64-91).  Here every scan of one
outer iteration is ONE kernel launch (f3d_points_in_obb*: all points x {box of id1, boxes of its candidates},
membership bitset in LDS, box-pair co-occurrence matrix); the control flow -- including the list-index-as-id
and delete-while-iterating quirks (Q6/Q7) and the early return at :83-84 -- stays on the host so ids match.

Oriented boxes: Open3D is optional.  With it, ``OrientedBoundingBox.create_from_points`` is used like the
reference; without it, ``obb_from_points`` below follows the same published recipe (convex hull -> PCA of the
hull vertices -> extents in that frame).  That fit is unpinned against Open3D.
"""
import json
import time
from pathlib import Path

import numpy as np

import f3d


def obb_from_points(pts):
    """(center [3], R [3,3] columns = axes, extent [3]) of the PCA-aligned box of the convex hull."""
    try:
        import open3d as o3d
        b = o3d.geometry.OrientedBoundingBox.create_from_points(o3d.utility.Vector3dVector(pts))
        return np.asarray(b.center), np.asarray(b.R), np.asarray(b.extent)
    except ImportError:
        pass
    from scipy.spatial import ConvexHull
    pts = np.asarray(pts, np.float64)
    hp = pts[np.sort(ConvexHull(pts).vertices)]
    mean = hp.mean(0)
    evals, evecs = np.linalg.eigh(np.cov((hp - mean).T, bias=True))
    R = evecs[:, np.argsort(-evals, kind='stable')]
    R[:, 2] = np.cross(R[:, 0], R[:, 1])
    loc = (hp - mean) @ R
    lo, hi = loc.min(0), loc.max(0)
    return mean + R @ ((lo + hi) / 2), R, hi - lo


def obb_corners(center, R, extent):
    x, y, z = (R[:, i] * extent[i] / 2 for i in range(3))
    c = np.asarray(center)
    return np.array([c - x - y - z, c + x - y - z, c - x + y - z, c - x - y + z,
                     c + x + y + z, c - x + y + z, c + x - y + z, c + x + y - z])


def _pack(boxes):
    return np.array([np.concatenate([c, np.asarray(R).reshape(-1), e]) for c, R, e in boxes], np.float64).reshape(-1, 15)


def intersection_point_bb(lst1, lst2):
    s = set(lst2)
    return [v for v in lst1 if v in s]


def update_id_info(id1, int_bb, info_sem, id_info_per_point):
    """Relabel instance int_bb as id1 and add its area (reference :58-62); arrays mutated in place."""
    info_sem[id1]["area"] += info_sem[int_bb]["area"]
    id_info_per_point[id_info_per_point == int_bb] = id1
    return info_sem, id_info_per_point


def check_intersection_open3d(id1, id_list, id_info_per_point, pcd_points, pcd, info_sem, box_fn=obb_from_points):
    """Ids (list indices) whose box shares a cloud point with the box of id1 (reference :68-91)."""
    own = pcd_points[id_info_per_point == id1]
    if len(own) < 4:
        return []
    boxes, cand = [box_fn(own)], []
    for id2 in range(1, len(id_list)):
        if id1 != id2 and id2 < len(info_sem) - 1 and id1 < len(info_sem) - 1:
            if info_sem[id1]["parent_id"] == info_sem[id2]["parent_id"]:
                other = pcd_points[id_info_per_point == id2]
                if len(other) < 4:
                    break                                  # the reference returns what it has so far (:83-84)
                boxes.append(box_fn(other))
                cand.append(id2)
    if not cand:
        return []
    hits = []
    for s in range(0, len(cand), f3d.MAX_OBB - 1):        # box 0 (id1) + up to 4095 candidates per launch
        part = [boxes[0]] + boxes[1 + s:1 + s + f3d.MAX_OBB - 1]
        _, cooc = f3d.default_context().points_in_obb(pcd_points, _pack(part), want_bits=False, want_cooc=True)
        hits += [cand[s + k] for k in np.nonzero(cooc[0, 1:])[0]]
    return hits


def merge_bb(dir_name, info_sem, id_info_per_point, pcd, box_fn=obb_from_points):
    """Merge same-parent instances whose boxes share a point; writes final_info.json and ids.npy (reference :103-137)."""
    n0 = len(info_sem)
    pts = np.ascontiguousarray(np.asarray(pcd.points if hasattr(pcd, 'points') else pcd), dtype=np.float64)
    t0 = time.perf_counter()
    id_list = [info_sem[i]["id"] for i in range(len(info_sem))]
    for id1 in range(1, len(id_list)):
        hits = check_intersection_open3d(id1, id_list, id_info_per_point, pts, pcd, info_sem, box_fn)
        if hits:
            for b in hits:
                info_sem, id_info_per_point = update_id_info(id1, b, info_sem, id_info_per_point)
            for b in hits:
                if b < len(info_sem):
                    del info_sem[b]
    for k in range(1, len(info_sem)):
        own = pts[id_info_per_point == info_sem[k]["id"]]
        if len(own) > 4:
            info_sem[k]["bbox"] = obb_corners(*box_fn(own)).tolist()
    print(f'Time taken for merging {n0} to {len(info_sem)} Bounding boxes = {time.perf_counter() - t0} seconds')
    if dir_name is not None:
        out = Path(dir_name) / "panoptic_segmentation"
        out.mkdir(parents=True, exist_ok=True)
        with open(out / "final_info.json", 'w') as fp:
            json.dump(info_sem, fp, indent=4)
        with open(out / "ids.npy", 'wb') as fi:
            np.save(fi, id_info_per_point)
    return info_sem, id_info_per_point
