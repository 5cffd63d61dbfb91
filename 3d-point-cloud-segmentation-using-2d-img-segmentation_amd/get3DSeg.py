"""Drop-in surface of the reference's get3DSeg.py: ``segment``, ``remove_classes``, ``master_classes`` and helpers.

Call order, defaults, return values and on-disk formats follow the reference (file:line cited per function); the
arithmetic runs on the GPU through the drop-in modules (VotingSegmentation -> f3d_vote_uv2pt / f3d_segment_votes,
merge_bb -> f3d_points_in_obb).  Open3D is optional: .ply files are written by a small binary writer here, the
interactive viewers of the reference (draw_geometries) are not reproduced.
"""
import json
import os
import time
from pathlib import Path

import numpy as np

from Fusion3DSeg.fusion import Fusion
from Fusion3DSeg.merge_intersecting_bb import merge_bb, obb_from_points, obb_corners
from Fusion3DSeg.segUtils.cv import split_into_instances
from Fusion3DSeg.segUtils.voting import VotingSegmentation

_HERE = Path(os.path.dirname(os.path.abspath(__file__)))
_COCO_META = _HERE.parent / 'deeplearning' / 'segmentation' / 'mask2former' / 'coco_meta.json'      # reference :68
_CLASSES_CSV = _HERE.parent / 'classes.csv'                                                           # reference :143,376
_CLASSES_META = _HERE.parent / 'classes_meta.json'                                                    # reference :377


class PointCloud:
    """Minimal stand-in for the o3d.geometry.PointCloud the reference passes around (only .points/.colors/.normals)."""

    def __init__(self, points, colors=None, normals=None):
        self.points, self.colors, self.normals = np.asarray(points, np.float64), colors, normals


def write_ply(path, pcd):
    """Binary little-endian PLY with x,y,z (double) and optional normals / uchar colours."""
    pts = np.asarray(pcd.points, np.float64)
    fields = [('x', '<f8'), ('y', '<f8'), ('z', '<f8')]
    cols = [pts[:, 0], pts[:, 1], pts[:, 2]]
    if pcd.normals is not None:
        nr = np.asarray(pcd.normals, np.float64)
        fields += [('nx', '<f8'), ('ny', '<f8'), ('nz', '<f8')]
        cols += [nr[:, 0], nr[:, 1], nr[:, 2]]
    if pcd.colors is not None:
        c8 = np.clip(np.asarray(pcd.colors) * 255.0, 0, 255).astype(np.uint8)
        fields += [('red', 'u1'), ('green', 'u1'), ('blue', 'u1')]
        cols += [c8[:, 0], c8[:, 1], c8[:, 2]]
    rec = np.empty(len(pts), dtype=fields)
    for (name, _), col in zip(fields, cols):
        rec[name] = col
    names = {'<f8': 'double', 'u1': 'uchar'}
    header = 'ply\nformat binary_little_endian 1.0\n' + f'element vertex {len(pts)}\n' + \
        ''.join(f'property {names[t]} {n}\n' for n, t in fields) + 'end_header\n'
    with open(path, 'wb') as fp:
        fp.write(header.encode('ascii'))
        fp.write(rec.tobytes())


def read_ply_points(path):
    """xyz of a PLY written by write_ply (or any binary-LE / ascii PLY whose first three properties are x,y,z)."""
    with open(path, 'rb') as fp:
        props, n, fmt = [], 0, 'ascii'
        while True:
            line = fp.readline().decode('ascii').strip()
            if line.startswith('format'):
                fmt = line.split()[1]
            elif line.startswith('element vertex'):
                n = int(line.split()[-1])
            elif line.startswith('property') and len(props) < 64 and 'list' not in line:
                props.append(line.split()[1:3])
            elif line == 'end_header':
                break
        np_t = {'double': '<f8', 'float': '<f4', 'uchar': 'u1', 'float64': '<f8', 'float32': '<f4', 'uint8': 'u1', 'int': '<i4'}
        if fmt == 'ascii':
            data = np.loadtxt(fp, max_rows=n, ndmin=2)
            return data[:, :3].astype(np.float64)
        rec = np.frombuffer(fp.read(), dtype=[(nm, np_t[t]) for t, nm in props], count=n)
        return np.stack([rec['x'], rec['y'], rec['z']], axis=1).astype(np.float64)


def _tocss(clr):
    return "#" + "".join(hex(int(c)).replace('0x', '').zfill(2) for c in clr)


def _coco_meta():
    return _COCO_META if _COCO_META.is_file() else None


def semantic_viz(points, classes, nclasses, votes=None, coco_data=None, outdir='./'):
    """classes.npy (+votes.npy), pcd.ply and info.json of the semantic result (reference :224-286)."""
    outdir = Path(outdir)
    outdir.mkdir(exist_ok=True, parents=True)
    if votes is not None:
        np.save(outdir / 'votes.npy', votes)
    np.save(outdir / 'classes.npy', classes)
    if coco_data is not None:
        with open(coco_data, 'r') as fp:
            names = list(json.load(fp)['stuff_classes'])
    else:
        names = [str(i) for i in range(nclasses)]
    names.append('unclassified')
    palette = np.vstack((np.random.uniform(0, 1, size=(nclasses, 3)), np.zeros((1, 3))))
    class_ids, counts = np.unique(classes, return_counts=True)
    colors = np.zeros_like(points)
    for c in class_ids:
        colors[classes == c, :] = palette[c]
    pcd = PointCloud(points, colors)
    write_ply(outdir / 'pcd.ply', pcd)
    hexes = [_tocss(c) for c in (palette * 255).astype(int)[class_ids]]
    info = [{'category_id': int(c), 'name': names[c], 'area': int(a), 'hexcolor': h} for c, a, h in zip(class_ids, counts, hexes)]
    with open(outdir / 'info.json', 'w') as fp:
        json.dump(info, fp, indent=4)
    return colors, pcd, hexes, info


def panoptic_viz(points, ids, idinfo, outdir, coco_data=None, colors=None, alpha=1.0):
    """ids.npy, info.json and pcd.ply of the panoptic result (reference :289-347)."""
    outdir = Path(outdir)
    outdir.mkdir(exist_ok=True, parents=True)
    np.save(outdir / 'ids.npy', ids)
    names = None
    if coco_data is not None:
        with open(coco_data, 'r') as fp:
            names = list(json.load(fp)['stuff_classes']) + ['unclassified']
    allids = np.unique(ids)
    idinfo = [idinfo[i] for i in allids]
    colors = np.zeros_like(points) if colors is None else colors
    palette = np.random.uniform(0, 1, size=(len(allids), 3))
    for i, info, clr in zip(allids, idinfo, palette):
        info['hexcolor'] = _tocss((clr * 255).astype(int))
        info['name'] = names[info['category_id']] if names is not None else str(info['category_id'])
        m = ids == i
        colors[m] = (1 - alpha) * colors[m] + alpha * clr
    with open(outdir / 'info.json', 'w') as fp:
        json.dump(idinfo, fp, indent=4)
    pcd = PointCloud(points, colors)
    write_ply(outdir / 'pcd.ply', pcd)
    return colors, pcd, palette, idinfo


def segment(dirname, mask_dir, threshold=0.5, nclasses=133, filter_classes=[86, 114, 115], min_pts_per_inst=100, verbose=True):
    """Semantic + panoptic segmentation of a fused cloud from 2D masks (reference :18-116)."""
    dirname = Path(dirname)
    points, norms, colors, nmerges, occurences, nframes, depth_hw, adj = Fusion.load_data(dirname)
    t0 = time.perf_counter()
    voter = VotingSegmentation(len(points), depth_hw, mask_dir, dirname / 'fusion' / 'uv2pt', nclasses, votes_file=None)
    votes = voter.vote(resize=True, filename=dirname / 'segmentation' / 'votes.npy', verbose=verbose)
    classes = voter.segment(threshold, filter_classes)
    if verbose:
        print(f'Time taken for segmentation = {time.perf_counter() - t0} seconds')
    if adj is not None:
        insts, ids, pan_info, pan_classes = split_into_instances(classes, adj, nclasses, filter_classes, min_pts_per_inst, verbose=verbose)
    else:
        print('No adjacency list available, hence skipping instance seperation.')
    semantic_viz(points, classes, nclasses, votes=None, coco_data=_coco_meta(), outdir=dirname / 'segmentation')
    if adj is None:
        return votes, classes
    panoptic_viz(points, ids, pan_info, dirname / 'panoptic_segmentation', _coco_meta(), colors=None, alpha=1.0)
    master_classes(dirname)


def remove_classes(dirname, mask_dir, keep_classes, threshold=0.75, nclasses=133, verbose=True):
    """Mask of points whose class survives (reference :118-221); reuses segmentation/votes.npy when present, in which
    case the voter's "unclassified" label is votes.shape[1] = 134 (quirk Q2), hence both 133 and 134 are removed."""
    _, _, _, _, keep_classes = load_csv(_CLASSES_CSV)
    dirname = Path(dirname)
    points, norms, colors, nmerges, occurences, nframes, depth_hw, adj = Fusion.load_data(dirname)
    colors_org = colors.copy()
    votes_file = dirname / 'segmentation' / 'votes.npy'
    votes_file = votes_file if votes_file.is_file() else None
    voter = VotingSegmentation(len(points), depth_hw, mask_dir, dirname / 'fusion' / 'uv2pt', nclasses, votes_file=votes_file)
    if votes_file is None:
        voter.vote(resize=True, filename=dirname / 'segmentation' / 'votes.npy', verbose=verbose)
    classes = voter.segment(threshold, None)
    removed = np.append(np.setdiff1d(np.arange(nclasses), keep_classes), [133, 134])
    remaining = ~np.isin(classes, removed)
    (dirname / 'segmentation').mkdir(exist_ok=True, parents=True)
    np.save(dirname / 'segmentation' / 'remaining_mask.npy', remaining)
    colors[remaining] = [1, 0, 0]
    colors[~remaining] = [0, 0, 1]
    write_ply(dirname / 'segmentation' / 'remaining.ply', PointCloud(points, colors))
    write_ply(dirname / 'segmentation' / 'cleaned.ply', PointCloud(points[remaining], colors_org[remaining], norms[remaining]))
    shown = classes.copy()
    shown[remaining] = 133
    shown[shown == 134] = 133
    semantic_viz(points, shown, nclasses, votes=None, coco_data=_coco_meta(), outdir=dirname / 'segmentation' / 'removed_objects_info')
    return remaining


def load_semantic_segmentation(semantic_dir):
    votes = np.load(os.path.join(semantic_dir, 'votes.npy'))
    classes = np.load(os.path.join(semantic_dir, 'classes.npy'))
    with open(os.path.join(semantic_dir, 'info.json'), 'r') as fp:
        info = json.load(fp)
    return votes, classes, classes, np.unique(classes), info


def load_csv(data_path):
    """class ids, parent names, parent ids, info-json flags, and the classes NOT flagged for removal (reference :357-367)."""
    import pandas as pd
    df = pd.read_csv(data_path)
    class_id = df['Class_ID'].tolist()
    keep = [class_id[i] for i in np.where(np.bool_(df['flag_objremoval'].tolist()) == False)[0]]  # noqa: E712
    return class_id, df['Parent'].tolist(), df['Parent_ID'].tolist(), df['flag_infojson'].tolist(), keep


def master_classes(dirname, classes_csv=None, meta_json=None):
    """Parent-class bookkeeping, per-instance oriented boxes, then merge_bb (reference :369-475)."""
    dirname = Path(dirname)
    class_id, parent_name, parent_id, flag_infojson, _ = load_csv(classes_csv or _CLASSES_CSV)
    with open(meta_json or _CLASSES_META, 'r') as fp:
        meta = json.load(fp)
    points = read_ply_points(dirname / 'panoptic_segmentation' / 'pcd.ply')
    ids = np.load(dirname / 'panoptic_segmentation' / 'ids.npy')
    classes = np.load(dirname / 'segmentation' / 'classes.npy')
    parent_classes = classes.copy()
    with open(dirname / 'panoptic_segmentation' / 'info.json', 'r') as fp:
        info_pan = json.load(fp)
    with open(dirname / 'segmentation' / 'info.json', 'r') as fp:
        info_sem = json.load(fp)
    palette = np.array(meta['colors']) / 255
    final_info, area_unclassified, unclassified_instance = [], 0, None
    for info in info_pan:
        if info['category_id'] in class_id:
            k = class_id.index(info['category_id'])
            info['parent_id'], info['parent_name'] = parent_id[k], parent_name[k]
            info['parent_hexcolor'] = _tocss((palette[info['parent_id']] * 255).astype(int))
            if info['category_id'] == 133:
                unclassified_instance, box = info['id'], None
            else:
                box = obb_corners(*obb_from_points(points[ids == info['id']])).tolist()
            info['bbox'] = box
            if flag_infojson[k]:
                final_info.append(info)
        else:
            area_unclassified += int(np.count_nonzero(ids == info['id']))
            info['parent_id'] = info['parent_name'] = info['parent_hexcolor'] = info['bbox'] = None
    final_info[unclassified_instance]['area'] += area_unclassified          # list index = instance id, as in the reference (:450)
    for info in info_sem:
        m = classes == info['category_id']
        if info['category_id'] in class_id:
            k = class_id.index(info['category_id'])
            info['parent_id'], info['parent_name'] = parent_id[k], parent_name[k]
            info['parent_hexcolor'] = _tocss((palette[info['parent_id']] * 255).astype(int))
            parent_classes[m] = int(info['parent_id'])
        else:
            parent_classes[m] = meta['classes'].index('unclassified')
    colors = np.zeros_like(points)
    for c in np.unique(parent_classes):
        colors[parent_classes == c] = palette[c]
    pcd = PointCloud(points, colors)
    write_ply(dirname / 'segmentation' / 'final_pcd.ply', pcd)
    with open(dirname / 'segmentation' / 'info.json', 'w') as fp:
        json.dump(info_sem, fp, indent=4)
    with open(dirname / 'panoptic_segmentation' / 'info.json', 'w') as fp:
        json.dump(info_pan, fp, indent=4)
    merge_bb(dirname, final_info, ids, pcd)
