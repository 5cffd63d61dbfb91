"""Drop-in surface of the reference's Fusion3DSeg/fusion.py for the hot path.

Provided here: ``parse_rts``, ``Fusion._get_frustum_data`` / ``filter`` / ``load_data`` (reference
fusion.py:67-77,119-132,329-347,389-407) and ``project_vote_argmax`` -- the forward
"project every point into every view, sample the mask, vote, segment" composition the north star names
(last row), which runs as ONE fused HIP kernel.

Not provided in this round: ``Fusion.fuse`` / ``patch_downsample`` (greedy, order-dependent patch merging with an
unseeded shuffle, reference :134-324) -- row (f)#2 of the scope table; calling them raises NotImplementedError.
"""
import pickle
from pathlib import Path

import numpy as np

import f3d


def parse_rts(rts):
    """Camera pickle -> (scaled intrinsics, depth w, depth h, wxyz quaternions, translations) (reference :67-77)."""
    with open(rts, 'rb') as fp:
        d = pickle.load(fp)
    h, w, *_ = d['Depth_res']
    return d['intrinsicScaled'], w, h, d['odo_wxyz'][:, [3, 0, 1, 2]], d['odo_xyz']


def project_vote_argmax(points, K, wxyzs, translations, masks, max_depth=10, nclasses=133, threshold=0.5,
                        filter_classes=None, return_votes=False):
    """Label every point from V posed masks in one pass on the GPU.

    Per view j (reference call sites): the 5 frustum planes of ``Fusion.fuse`` (fusion.py:254-258) ->
    ``point_inside_polyhedra`` (:260) -> ``points2pixel`` (:266) -> samples outside the mask are dropped ->
    ``votes[point, mask_j[v, u]] += 1`` -> ``VotingSegmentation.segment(threshold, filter_classes)``.

    points [N,3] float64 (or float32), K [3,3], wxyzs [V,4] camera->world (w,x,y,z), translations [V,3],
    masks uint8 [V,H,W] (H, W also define the frustum).  Returns int64 [N] (and uint16 [N, nclasses+1] votes).
    """
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    V, H, W = masks.shape
    views = f3d.views_build(K, W, H, wxyzs, translations, max_depth)
    return f3d.default_context().project_vote_argmax(points, views, masks, nclasses, threshold, filter_classes, return_votes)


def radius_adjacency(points, ds_radius, as_csr=False):
    """The adjacency ``Fusion.save_data`` stores in fusion/adj.pkl (reference :373-377):
    ``KDTree(points).query_radius(points, r=2 * ds_radius)``, built on the GPU with a uniform grid.

    Returns what the reference pickles -- an object array of int64 index arrays, one per point, itself included -- or,
    with ``as_csr``, the (offsets int64 [N+1], neighbours int32 [E]) pair that ``split_into_instances`` also accepts
    (no Python objects, what a 10M-point cloud wants).  Row order: by grid cell, then ascending index (sklearn's order is
    the tree traversal's, unspecified)."""
    if ds_radius is None:
        return None                                                            # reference :371-372
    offs, nbrs = f3d.default_context().radius_graph(points, 2 * ds_radius)
    if as_csr:
        return offs, nbrs
    out = np.empty(len(offs) - 1, dtype=object)
    wide = nbrs.astype(np.int64)
    for i in range(len(out)):
        out[i] = wide[offs[i]:offs[i + 1]]
    return out


class Fusion:
    def __init__(self, tof, rts, point_range=None, decimation=1, save_lookups=True):
        raise NotImplementedError('Fusion.fuse (greedy patch merge, reference fusion.py:134-324) is outside this round\'s scope; '
                                  'use Fusion.load_data on an existing fusion directory, or fusion.project_vote_argmax')

    @staticmethod
    def _get_frustum_data(K, w, h, xyzws, translations, frame_ids=None):
        """eyes [F,3], lookats [F,3], spoke origins [F,4,3], face normals [F,4,3] (reference :119-132), computed by the
        library's host code.  With ``frame_ids`` the reference indexes the eyes twice for the spoke origins; kept."""
        eyes, lookats, normals = f3d.frustum_data(K, w, h, xyzws, translations)
        ids = np.arange(len(eyes)) if frame_ids is None else np.asarray(frame_ids)
        eyes, lookats = eyes[ids], lookats[ids]
        return eyes, lookats, np.repeat(eyes[ids][:, None, :], 4, axis=1), normals[ids]

    @staticmethod
    def filter(values, threshold, data=None, less_than=False):
        mask = values <= threshold if less_than else values >= threshold
        return (mask, None) if data is None else (mask, [d[mask] for d in data])

    @classmethod
    def load_data(cls, dirname):
        """points, normals, colors, nmerges, occurences, nframes, depth_hw, adj (reference :389-407)."""
        dirname = Path(dirname)
        with open(dirname / 'fusion' / 'fusion_data.pkl', 'rb') as fp:
            d = pickle.load(fp)
        adj = None
        if (dirname / 'fusion' / 'adj.pkl').is_file():
            with open(dirname / 'fusion' / 'adj.pkl', 'rb') as fp:
                adj = pickle.load(fp)
        return [d['points'], d['normals'], d['colors'], d['nmerges'], d['occurences'], d['nframes'], d['depth_hw'], adj]
