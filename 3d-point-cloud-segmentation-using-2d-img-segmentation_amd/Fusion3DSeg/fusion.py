"""Drop-in surface of the reference's Fusion3DSeg/fusion.py for the hot path.

``project_vote_argmax`` is the forward "project every point into every view, sample the mask, vote, segment" composition
the north star names (last row); it runs as ONE fused HIP kernel.

``Fusion`` is the reference's class (fusion.py:80-407): per frame, ``fuse`` culls the fused cloud against the frame's
frustum and projects the survivors (rows a3, a4, a2 -- here ONE launch of the single-view HIP kernel per frame), then
greedily matches depth patches to those points and down-samples what is left (``patch_downsample``).  The greedy part is
a chain of data-dependent decisions (every match removes pixels from later candidates) and runs on the host, as in the
reference; it draws its shuffles from NumPy's global generator at the same places, so a seeded run reproduces the
reference's output (tests/golden/fuse.npz).
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


class FrameData:
    """Per-frame pickles listed by the tof pickle (reference :17-60): item i -> (frame name, world points [h*w,3], normals,
    colours, validity mask)."""

    def __init__(self, tof, point_range=None, decimation=1, depth_hw=(256, 192)):
        self.point_range, self.decimation, self.depth_hw = point_range, decimation, depth_hw
        root = Path(str(tof).split('PointcloudMergeResults')[0])
        with open(tof, 'rb') as fp:
            self.tofcamedata = [root / entry['fileName'].strip() for entry in pickle.load(fp)]

    def __len__(self):
        return len(self.tofcamedata)

    @staticmethod
    def get_valid(points, mindist, maxdist):
        depth = points[:, 2]
        return (depth > mindist) & (depth <= maxdist)

    def __getitem__(self, i):
        with open(self.tofcamedata[i], 'rb') as fp:
            d = pickle.load(fp)
        pts = np.array(d['modPoints'])
        if self.point_range is None:
            ok = np.ones(len(pts), bool)
        else:
            ok = self.get_valid(np.array(d['orgPoints']), self.point_range[0], self.point_range[1])
        if self.decimation > 1:                                               # keep one pixel per decimation x decimation block
            drop = np.ones(self.depth_hw, bool)
            drop[::self.decimation, ::self.decimation] = False
            ok[drop.reshape(-1)] = False
        return str(d['frameNumber']), pts, np.array(d['modSurfaceNormals']), np.array(d['orgColorPoints']), ok


def _row_norms(rows):
    """np.linalg.norm(v) of every row, with the bits of the reference's per-vector call: for a 1-D vector NumPy takes
    sqrt(v.dot(v)), and that dot is BLAS ddot (FMA, kernel dependent), unlike the axis=-1 form.  ``np.vecdot`` runs the same
    kernel on the builds seen so far; it is used for large inputs only after it reproduced the per-vector results on the first
    rows of THIS input."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    head = min(len(rows), 1024)
    sq = np.empty(len(rows))
    sq[:head] = [v.dot(v) for v in rows[:head]]
    if head < len(rows):
        fast = getattr(np, 'vecdot', None)
        if fast is not None and np.array_equal(fast(rows[:head], rows[:head]), sq[:head]):
            sq[head:] = fast(rows[head:], rows[head:])
        else:
            sq[head:] = [v.dot(v) for v in rows[head:]]
    return np.sqrt(sq)


def _mergeable(seed_pt, seed_normal, cand_pts, cand_normals, max_distance, min_cosine):
    """The reference's merge criterion (:165-170, :223-228): closer than max_distance AND normals within the angle."""
    near = np.linalg.norm(cand_pts - seed_pt[None, :], axis=-1) < max_distance
    return near & (np.einsum('ij, j -> i', cand_normals, seed_normal) > min_cosine)


class Fusion:
    def __init__(self, tof, rts, point_range=None, decimation=1, save_lookups=True):
        K, w, h, wxyzs, translations = parse_rts(rts)
        lookup_dir = None
        if save_lookups:
            lookup_dir = Path(str(tof).split('PointcloudMergeResults')[0]) / 'fusion' / 'uv2pt'
            lookup_dir.mkdir(exist_ok=True, parents=True)
        self._setup(K, w, h, wxyzs, translations, FrameData(tof, point_range, decimation, (h, w)), save_lookups, lookup_dir)

    @classmethod
    def from_frames(cls, K, w, h, wxyzs, translations, frames, lookup_dir=None, lookup_sink=None):
        """The same object without the file readers: ``frames[i]`` = (name, points, normals, colours, valid).  Per-frame
        lookups go to ``lookup_dir`` (as .npy, like the reference) and/or to ``lookup_sink(name, uv2pt)``."""
        self = object.__new__(cls)
        self._setup(K, w, h, wxyzs, translations, frames, lookup_dir is not None or lookup_sink is not None, lookup_dir)
        self._lookup_sink = lookup_sink
        return self

    def _setup(self, K, w, h, wxyzs, translations, frames, save_lookups, lookup_dir):
        self.K, self.w, self.h = np.asarray(K, np.float64), int(w), int(h)
        self.xyzws, self.translations = np.asarray(wxyzs, np.float64), np.asarray(translations, np.float64)   # (w,x,y,z): quirk Q9
        self.frames, self.nframes, self.npts = frames, len(frames), int(h) * int(w)
        self.ds_radius, self.ds_angle = None, None
        self.eyes, self.lookats, self.frustum_spoke_origins, self.frutsum_face_normals = self._get_frustum_data(
            self.K, self.w, self.h, self.xyzws, self.translations, np.arange(self.nframes))
        self.pcdimg = np.arange(self.npts).reshape(self.h, self.w)
        self.pt2u = (np.arange(self.npts) % self.w).astype(np.int32)
        self.pt2v = (np.arange(self.npts) // self.w).astype(np.int32)
        self.save_lookups, self.uv2pt_dir, self._lookup_sink = save_lookups, lookup_dir, None

    @staticmethod
    def _get_frustum_data(K, w, h, xyzws, translations, frame_ids=None):
        """eyes [F,3], lookats [F,3], spoke origins [F,4,3], face normals [F,4,3] (reference :119-132), computed by the
        library's host code.  With ``frame_ids`` the reference indexes the eyes twice for the spoke origins; kept."""
        eyes, lookats, normals = f3d.frustum_data(K, w, h, xyzws, translations)
        ids = np.arange(len(eyes)) if frame_ids is None else np.asarray(frame_ids)
        eyes, lookats = eyes[ids], lookats[ids]
        return eyes, lookats, np.repeat(eyes[ids][:, None, :], 4, axis=1), normals[ids]

    @classmethod
    def patch_downsample(cls, points, normals, colors, height, width, stride, max_distance, min_cosine,
                         pcdimg, pt2u, pt2v, non_merged=None):
        """One frame -> representative points (reference :134-210): visit the pixels in a random order; a pixel that is still
        free becomes a seed, absorbs the free pixels of its (stride x stride) window that satisfy the merge criterion and
        is replaced by their mean.  Returns (points, normals, colours, uv2pt int32 [h*w] with -1 = none, merge counts).

        "Still free when visited" is the only sequential coupling, and it is local: pixel p is a seed iff no EARLIER seed whose
        window covers it accepts it.  The HIP kernels settle that in a few data-parallel rounds and return, per pixel, the seed
        that takes it; the host then only adds the members of every seed in the reference's order."""
        order = np.arange(len(points))
        np.random.shuffle(order)                                              # the global generator, as the reference (:172)
        free = np.ones((height, width), dtype=bool) if non_merged is None else non_merged   # updated in place, like the reference
        half = stride // 2
        points, normals = np.asarray(points, np.float64), np.asarray(normals, np.float64)
        flat_free = free.reshape(-1)
        own_cos = np.einsum('ij,ij->i', normals, normals)
        usable = (own_cos > min_cosine) & np.isfinite(points).all(axis=1)
        if len(points) != height * width or not (max_distance > 0) or (flat_free & ~usable).any():
            # a free pixel that would not accept itself (zero / NaN normal): the reference then averages an empty set and leaves the
            # pixel to later seeds -- keep its literal order of events for such frames
            return cls._patch_downsample_sequential(order, points, normals, colors, height, width, half, max_distance, min_cosine,
                                                    pcdimg, pt2u, pt2v, free)
        prio = np.empty(len(points), np.int32)
        prio[order] = np.arange(len(points), dtype=np.int32)
        # seeds, what each of them takes, and the ordered sums of those rows (a thread per seed adds its members in ascending pixel
        # index = the reference's window order, so the means are bit-identical): one call, the frame uploaded once
        colors = np.asarray(colors, np.float64)
        owner, sums, counts, _ = f3d.default_context().patch_seeds_sums(points, normals, colors, prio, flat_free, height, width, half,
                                                                        max_distance, min_cosine)
        uv2pt = np.full(height * width, -1, np.int32)
        seeds = np.nonzero(owner == np.arange(height * width, dtype=np.int32))[0]
        if not len(seeds):
            return np.array([]), np.array([]), np.array([]), uv2pt, np.array([])
        seeds = seeds[np.argsort(prio[seeds], kind='stable')]                # seeds in visiting order = output order
        n_take = counts[seeds].astype(np.int64)
        mean = sums[seeds] / n_take[:, None]                                 # np.mean: the ordered sum, then one division
        out_p, nsum, out_c = mean[:, 0:3], mean[:, 3:6], mean[:, 6:9]
        out_n = nsum / _row_norms(nsum)[:, None]
        rank = np.empty(height * width, np.int32)
        rank[seeds] = np.arange(len(seeds), dtype=np.int32)
        taken = owner >= 0
        uv2pt[taken] = rank[owner[taken]]
        free.reshape(-1)[taken] = False
        return np.ascontiguousarray(out_p), out_n, np.ascontiguousarray(out_c), uv2pt, n_take

    @staticmethod
    def _patch_downsample_sequential(order, points, normals, colors, height, width, half, max_distance, min_cosine, pcdimg, pt2u, pt2v, free):
        """The literal order of events of reference :176-208 (used only for frames the data-parallel form does not cover)."""
        uv2pt = np.full(height * width, -1, np.int32)
        left = height * width
        out_p, out_n, out_c, out_m = [], [], [], []
        for seed in order:
            su, sv = pt2u[seed], pt2v[seed]
            if not free[sv, su]:
                continue
            if not left:
                break
            win = np.s_[max(0, sv - half):sv + half + 1, max(0, su - half):su + half + 1]
            cand = pcdimg[win].reshape(-1)[free[win].reshape(-1)]           # row-major window order, free pixels only
            take = _mergeable(points[seed], normals[seed], points[cand], normals[cand], max_distance, min_cosine)
            members = cand[take]
            left -= take.sum()
            out_p.append(np.mean(points[cand][take], axis=0))
            out_c.append(np.mean(colors[cand][take], axis=0))
            nsum = np.mean(normals[cand][take], axis=0)
            out_n.append(nsum / np.linalg.norm(nsum))
            out_m.append(take.sum())
            uv2pt[members] = len(out_m) - 1
            free[pt2v[members], pt2u[members]] = False
        return np.array(out_p), np.array(out_n), np.array(out_c), uv2pt, np.array(out_m)

    def _frame_view(self, j, max_depth):
        return f3d.views_build(self.K, self.w, self.h, self.xyzws[j:j + 1], self.translations[j:j + 1], max_depth)[0]

    def fuse(self, radius=0.05, angle=10, stride=None, max_depth=10, skip=1, verbose=False):
        """Fuse + down-sample the frames into one sparse cloud (reference :212-324) ->
        (points, normals, colours, nmerges, occurences); per-frame ``uv2pt`` lookups are saved when requested."""
        self.ds_radius, self.ds_angle = radius, angle
        stride = max(10, int(radius * 200)) if stride is None else stride
        half, min_cosine = stride // 2, np.cos(np.deg2rad(angle))
        ctx = f3d.default_context()
        for first in range(0, self.nframes):                                 # first frame with any valid pixel seeds the cloud
            name, pts, nrm, clr, valid = self.frames[first]
            if valid.any():
                break
        pts, nrm, clr, uv2pt, nmerges = self.patch_downsample(pts, nrm, clr, self.h, self.w, stride, radius, min_cosine,
                                                               self.pcdimg, self.pt2u, self.pt2v, valid.reshape(self.h, self.w))
        if self.save_lookups:
            self._save_uv2pt(uv2pt, name)
        occurences = np.ones(len(pts), np.uint32)
        hits = np.ones(self.npts, dtype=bool)
        for j in range(first + 1, self.nframes, skip):
            if verbose:
                print(f'fusing frame: {j + 1}, total points = {len(pts)}, previous intersections = {hits.sum()}')
            name, q_pts, q_nrm, q_clr, q_valid = self.frames[j]
            if not q_valid.any():
                continue
            uv2pt = np.full(self.npts, -1, np.int32)
            # a3 + a4 + a2 of the reference (:254-266) in one launch: frustum (4 sides + far plane at max_depth) membership of
            # every fused point and the pixel it projects to
            uv_all, hits = ctx.project_view(pts, self._frame_view(j, max_depth))
            if hits.any():
                ids = np.where(hits)[0]
                x_pts, x_nrm, x_clr = pts[hits], nrm[hits], clr[hits]
                x_mrg, x_occ = nmerges[hits], occurences[hits]
                uv = uv_all[:, hits]
                free = q_valid.reshape(self.h, self.w)                      # a view: the frame's mask is consumed, as in the reference
                # The reference visits the seeds in index order and lets each take the free pixels of its window that pass the
                # criterion (:269-298).  A seed's criterion uses its own position / normal from BEFORE this frame, so the result
                # is: a free pixel belongs to the first seed whose window covers it and accepts it -- one HIP launch.
                # ... followed, on the resident frame, by one thread per seed that adds the rows of the pixels it took in ascending
                # pixel index (((r0 + r1) + r2) + ..., the order np.mean adds the reference's vstack in)
                owner, sums, counts = ctx.patch_match(uv, x_pts, x_nrm, q_pts, q_nrm, q_clr, free.reshape(-1), self.h, self.w, half, radius,
                                                      min_cosine)
                seeds = np.nonzero(counts)[0]
                if len(seeds):
                    n_take = counts[seeds].astype(np.int64)
                    denom = (n_take + 1)[:, None]                           # the seed itself is the last row of the reference's vstack
                    x_pts[seeds] = (sums[seeds, 0:3] + x_pts[seeds]) / denom
                    x_clr[seeds] = (sums[seeds, 6:9] + x_clr[seeds]) / denom
                    nsum = (sums[seeds, 3:6] + x_nrm[seeds]) / denom
                    x_nrm[seeds] = nsum / _row_norms(nsum)[:, None]
                    x_mrg[seeds] += n_take
                    x_occ[seeds] += 1
                    taken = owner >= 0
                    uv2pt[taken] = ids[owner[taken]]
                    free.reshape(-1)[taken] = False
                pts[hits], nrm[hits], clr[hits] = x_pts, x_nrm, x_clr
                nmerges[hits], occurences[hits] = x_mrg, x_occ
            if free.any():                                                    # (as in the reference, `free` of an earlier frame if none hit)
                n_pts, n_nrm, n_clr, n_uv2pt, n_mrg = self.patch_downsample(q_pts, q_nrm, q_clr, self.h, self.w, 2 * stride, radius,
                                                                             min_cosine, self.pcdimg, self.pt2u, self.pt2v, free)
                fresh = n_uv2pt != -1
                uv2pt[fresh] = n_uv2pt[fresh] + len(pts)
                pts, nrm, clr = np.vstack([pts, n_pts]), np.vstack([nrm, n_nrm]), np.vstack([clr, n_clr])
                nmerges = np.hstack([nmerges, n_mrg])
                occurences = np.hstack([occurences, np.ones(len(n_pts), np.uint32)])
            if self.save_lookups:
                self._save_uv2pt(uv2pt, name)
        return pts, nrm, clr, nmerges, occurences

    def _save_uv2pt(self, uv2pt, frame_name):
        if self.uv2pt_dir is not None:
            np.save(Path(self.uv2pt_dir) / f'{frame_name}.npy', uv2pt)
        if self._lookup_sink is not None:
            self._lookup_sink(frame_name, uv2pt)

    @staticmethod
    def filter(values, threshold, data=None, less_than=False):
        mask = values <= threshold if less_than else values >= threshold
        return (mask, None) if data is None else (mask, [d[mask] for d in data])

    def dump_data(self, dirname, points, normals=None, colors=None, nmerges=None, occurences=None, compute_adjacency=True,
                  verbose=False):
        """fusion/fusion_data.pkl, fusion/adj.pkl and the .ply of the fused cloud (reference :349-387); the adjacency is the
        GPU radius graph (``radius_adjacency``) instead of sklearn's KD-tree."""
        dirname = Path(dirname)
        (dirname / 'fusion').mkdir(exist_ok=True, parents=True)
        if verbose:
            print(f'writing fusion data into "{dirname}" directory')
        record = {'points': points, 'normals': normals, 'colors': colors, 'nmerges': nmerges, 'occurences': occurences,
                  'nframes': self.nframes, 'depth_hw': (self.h, self.w)}
        with (dirname / 'fusion' / 'fusion_data.pkl').open('wb') as fp:
            pickle.dump(record, fp)
        if compute_adjacency:
            if verbose:
                print('computing adjacency ...')
            adj = radius_adjacency(points, self.ds_radius)
            with (dirname / 'fusion' / 'adj.pkl').open('wb') as fp:
                pickle.dump(None if adj is None else np.array(adj, dtype=object), fp)
        from get3DSeg import PointCloud, write_ply
        tag = str(self.ds_radius).replace('.', '_')
        write_ply(dirname / 'fusion' / f'fusion_{tag}_{self.ds_angle}.ply', PointCloud(points, colors, normals))

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
