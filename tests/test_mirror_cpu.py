"""Host-side logic of the drop-in modules (no GPU needed)."""
import json

import numpy as np
import pytest

from Fusion3DSeg.segUtils.voting import VotingSegmentation, resize_nearest
from Fusion3DSeg.merge_intersecting_bb import obb_from_points, obb_corners
import get3DSeg
from oracle import np_ref as O


def test_oracle_split_into_instances_matches_reference_golden(golden):
    g = golden('split_instances')
    offs, flat = g['adj_offsets'], g['adj_flat']
    adj = [flat[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    for i in range(int(g['ncases'])):
        ic = g[f'case{i}_instance_classes'].tolist() if g[f'case{i}_has_instance_classes'] else None
        insts, ids, info, newcls = O.split_into_instances(g['classes'], adj, 133, ic, int(g[f'case{i}_minimum_points']))
        assert len(insts) == int(g[f'case{i}_ninst'])
        assert np.array_equal(ids, g[f'case{i}_ids']) and np.array_equal(newcls, g[f'case{i}_classes'])
        got = np.array([[d['id'], int(d['isthing']), d['category_id'], d['area']] for d in info], np.int64).reshape(-1, 4)
        assert np.array_equal(got, g[f'case{i}_info'])


def test_resize_nearest_index_rule():
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert np.array_equal(resize_nearest(m, 4, 3), m)
    up = resize_nearest(m, 8, 6)
    assert up.shape == (6, 8) and np.array_equal(up[::2, ::2], m)
    down = resize_nearest(np.arange(64, dtype=np.uint8).reshape(8, 8), 4, 4)
    assert np.array_equal(down, np.arange(64).reshape(8, 8)[::2, ::2])


def test_voting_constructor_pairs_frames_by_stem_and_q2(tmp_path):
    from PIL import Image
    md, ud = tmp_path / 'masks', tmp_path / 'uv2pt'
    md.mkdir(); ud.mkdir()
    for stem in ('a', 'b', 'c'):
        Image.fromarray(np.zeros((4, 4), np.uint8)).save(md / f'{stem}.png')
    for stem in ('b', 'c', 'd'):
        np.save(ud / f'{stem}.npy', np.full(16, -1, np.int32))
    v = VotingSegmentation(10, (4, 4), md, ud, 133)
    assert v.votes.shape == (10, 134) and v.votes.dtype == np.float64 and v.nframes == 2
    assert sorted(p.stem for p in v.mask_files) == ['b', 'c'] == sorted(p.stem for p in v.uv2pt_files)
    np.save(tmp_path / 'votes.npy', np.ones((10, 134)))
    v2 = VotingSegmentation(10, (4, 4), md, ud, 133, votes_file=tmp_path / 'votes.npy')
    assert v2.nclasses == 134                                       # quirk Q2


def test_ply_round_trip(tmp_path):
    pts = np.random.default_rng(0).normal(size=(50, 3))
    get3DSeg.write_ply(tmp_path / 'a.ply', get3DSeg.PointCloud(pts, np.random.default_rng(1).random((50, 3)), pts))
    assert np.array_equal(get3DSeg.read_ply_points(tmp_path / 'a.ply'), pts)


def test_obb_fit_contains_its_points_and_matches_oracle_recipe():
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(300, 3)) * [2.0, 0.5, 0.1] @ np.linalg.qr(rng.normal(size=(3, 3)))[0].T + [1, 2, 3]
    c, R, e = obb_from_points(pts)
    assert O.points_in_obb(pts, c, R, e * (1 + 1e-12)).all()
    c2, R2, e2 = O.obb_from_points(pts)
    assert np.allclose(c, c2) and np.allclose(e, e2) and np.allclose(np.abs(R), np.abs(R2))
    assert obb_corners(c, R, e).shape == (8, 3)
    assert e[0] >= e[1] >= e[2]


def test_dead_aabb_overlap_pair_is_importable_and_keeps_its_quirks():
    """cal_min_max / check_intersection (reference merge_intersecting_bb.py:15-56, dead code; skspatial absent -> unpinned):
    axis extents of the box corners, the zero-dropping filter, and the result list that is reset inside the loop."""
    from Fusion3DSeg.merge_intersecting_bb import cal_min_max, check_intersection, obb_corners, obb_from_points
    rng = np.random.default_rng(4)
    pts = np.vstack([rng.normal(size=(200, 3)) * [1.0, 0.4, 0.2] + c for c in ([3.0, 2.0, 1.0], [3.5, 2.2, 1.1], [30.0, 2.0, 1.0])])
    ids = np.repeat([1, 2, 3], 200)
    mn_x, mx_x, mn_y, mx_y, mn_z, mx_z = cal_min_max(1, ids, pts)
    corners = obb_corners(*obb_from_points(pts[:200]))
    assert mn_x.shape == (1,) and mn_x[0] == corners[:, 0].min() and mx_z[0] == corners[:, 2].max()
    info = [{'category_id': 86}] * 4
    id_list = [0, 1, 2, 3]
    assert check_intersection(1, id_list, ids, pts, info) == []           # id2 = 3 is the last one looked at: far away
    assert check_intersection(1, id_list[:3], ids, pts, info) == [2]      # now id2 = 2 is the last: boxes overlap
    info2 = [{'category_id': 86}, {'category_id': 86}, {'category_id': 3}]
    assert check_intersection(1, id_list[:3], ids, pts, info2) == []      # other category


def test_load_csv(tmp_path):
    (tmp_path / 'classes.csv').write_text('Class_ID,Parent,Parent_ID,flag_infojson,flag_objremoval\n86,wall,1,1,0\n114,floor,2,1,0\n3,car,5,0,1\n')
    cid, pname, pid, fj, keep = get3DSeg.load_csv(tmp_path / 'classes.csv')
    assert cid == [86, 114, 3] and pid == [1, 2, 5] and keep == [86, 114] and pname[0] == 'wall'


def _write_capture(root, g, nframes):
    """The on-disk layout Fusion(tof, rts) reads (reference fusion.py:17-47,67-77): a tof pickle listing per-frame pickles, an rts
    pickle with the scaled intrinsics, depth resolution and (x,y,z,w) poses."""
    import pickle
    h, w = (int(x) for x in g['hw'])
    merged = root / 'PointcloudMergeResults'
    (merged / 'frames').mkdir(parents=True)
    entries = []
    for j in range(nframes):
        rel = f'PointcloudMergeResults/frames/f{j}.pkl'
        cam_z = np.full(h * w, 1.0); cam_z[~g['valid'][j]] = 0.0            # orgPoints only feed the range test on z
        with open(root / rel, 'wb') as fp:
            pickle.dump({'frameNumber': 100 + j, 'orgPoints': np.stack([cam_z * 0, cam_z * 0, cam_z], 1), 'modPoints': g['points'][j],
                         'modSurfaceNormals': g['normals'][j], 'orgColorPoints': g['colors'][j]}, fp)
        entries.append({'fileName': rel + ' \n'})
    with open(merged / 'tofsegment_demo_1.pkl', 'wb') as fp:
        pickle.dump(entries, fp)
    with open(merged / 'rtscameradata_demo_1.pkl', 'wb') as fp:
        pickle.dump({'Depth_res': (h, w), 'intrinsicScaled': g['K'], 'odo_wxyz': g['wxyz'][:, [1, 2, 3, 0]], 'odo_xyz': g['t']}, fp)
    return merged


def test_patch_downsample_and_frame_reader_match_reference_golden(golden, tmp_path):
    """Host-side half of rows a5 / (f)#2 (no GPU): the first frame's lookup of the reference run is patch_downsample alone
    (fusion.py:134-210, first draw of the seeded global generator), and FrameData reads the capture layout."""
    from Fusion3DSeg.fusion import FrameData, Fusion, parse_rts
    g = golden('fuse')
    h, w = (int(x) for x in g['hw'])
    np.random.seed(int(g['c0_params'][5]))
    pcdimg = np.arange(h * w).reshape(h, w)
    pt2u, pt2v = (np.arange(h * w) % w).astype(np.int32), (np.arange(h * w) // w).astype(np.int32)
    order = np.arange(h * w)
    np.random.shuffle(order)                 # the sequential form (the public patch_downsample runs the HIP kernels: GPU tests)
    pts, nrm, clr, uv2pt, nmerges = Fusion._patch_downsample_sequential(order, g['points'][0], g['normals'][0], g['colors'][0], h, w, 5, 0.05,
                                                                        np.cos(np.deg2rad(10)), pcdimg, pt2u, pt2v,
                                                                        g['valid'][0].copy().reshape(h, w))
    assert uv2pt.dtype == np.int32 and np.array_equal(uv2pt, g['c0_uv2pt'][0])
    assert len(pts) == len(nrm) == len(clr) == len(nmerges) == uv2pt.max() + 1 and nmerges.sum() == (uv2pt >= 0).sum()
    merged = _write_capture(tmp_path, g, 2)
    K, rw, rh, wxyz, t = parse_rts(merged / 'rtscameradata_demo_1.pkl')
    assert (rw, rh) == (w, h) and np.array_equal(wxyz, g['wxyz']) and np.array_equal(K, g['K'])
    fd = FrameData(str(merged / 'tofsegment_demo_1.pkl'), point_range=(0.1, 4), decimation=1, depth_hw=(h, w))
    name, p1, n1, c1, ok = fd[1]
    assert len(fd) == 2 and name == '101' and np.array_equal(p1, g['points'][1]) and np.array_equal(ok, g['valid'][1])
    ok2 = FrameData(str(merged / 'tofsegment_demo_1.pkl'), None, 2, (h, w))[0][4]
    assert ok2.reshape(h, w)[::2, ::2].all() and ok2.sum() == (h // 2) * (w // 2)


def test_oneformer_wrapper_calls_the_oneformer_predictor_with_task_semantic(tmp_path, monkeypatch):
    """get2DSeg.py:12,35,77: the predictor is OneFormer's own ``demo.defaults.DefaultPredictor`` (found through the ./OneFormer
    path entry), called as ``predictor(image, task='semantic')``; detectron2's engine predictor takes the image only.  The
    third-party packages are absent from the image: test-local fakes on sys.path record the calls (nothing is installed)."""
    import importlib
    import sys
    root = tmp_path / 'fake_site'
    for pkg, body in {
        'detectron2/__init__.py': '',
        'detectron2/config.py': 'class _Cfg:\n    def __init__(self):\n        self.MODEL = type("M", (), {})()\n        self.merged = None\n'
                                '    def merge_from_file(self, f):\n        self.merged = f\n\ndef get_cfg():\n    return _Cfg()\n',
        'detectron2/projects/__init__.py': '',
        'detectron2/projects/deeplab.py': 'def add_deeplab_config(cfg):\n    cfg.order = ["deeplab"]\n',
        'detectron2/engine/__init__.py': '',
        'detectron2/engine/defaults.py': 'class DefaultPredictor:\n    def __init__(self, cfg):\n        raise AssertionError("the reference uses '
                                         'demo.defaults.DefaultPredictor, not detectron2\'s")\n',
        'oneformer/__init__.py': ''.join(f'def add_{n}_config(cfg):\n    cfg.order.append("{n}")\n\n' for n in ('oneformer', 'common', 'swin', 'dinat', 'convnext')),
        'demo/__init__.py': '',
        'demo/defaults.py': 'calls = []\n\nclass DefaultPredictor:\n    def __init__(self, cfg):\n        self.cfg = cfg\n'
                            '    def __call__(self, image, task):\n        calls.append((image.shape, task))\n        return {"sem_seg": "logits", "panoptic_seg": None, "instances": None}\n',
    }.items():
        path = root / pkg
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(body)
    monkeypatch.syspath_prepend(str(root))
    for name in [m for m in sys.modules if m.split('.')[0] in ('detectron2', 'oneformer', 'demo')]:
        monkeypatch.delitem(sys.modules, name)
    import get2DSeg
    importlib.invalidate_caches()
    model = get2DSeg.OneFormer()
    out = model.predict(np.zeros((6, 8, 3), np.uint8))
    import demo.defaults as dd
    assert dd.calls == [((6, 8, 3), 'semantic')] and out['sem_seg'] == 'logits'
    assert type(model.predictor) is dd.DefaultPredictor
    assert model.predictor.cfg.order == ['deeplab', 'common', 'swin', 'dinat', 'convnext', 'oneformer']        # :46-52
    assert model.predictor.cfg.merged.endswith('oneformer_swin_large_bs16_100ep.yaml')
    assert model.predictor.cfg.MODEL.WEIGHTS.endswith('150_16_swin_l_oneformer_coco_100ep.pth')
    for name in [m for m in sys.modules if m.split('.')[0] in ('detectron2', 'oneformer', 'demo')]:
        monkeypatch.delitem(sys.modules, name)


def test_get2dseg_seeds_like_the_reference():
    """get2DSeg.py:83-89: every generator is seeded with 0 before the loop."""
    import random
    import torch
    import get2DSeg
    get2DSeg._seed_everything(0)
    a = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    assert a == (random.random(), float(np.random.rand()), float(torch.rand(1)))
    assert torch.backends.cudnn.deterministic is True and torch.backends.cudnn.benchmark is False


class _NumpyCloud:
    """The HIP cloud's interface on a machine without a GPU (grouping by argsort, the oracle's in-box test)."""

    def __init__(self, pts, share):
        self.pts, (self.lo, self.hi) = pts, share

    def group(self, ids, nids):
        order = np.argsort(ids, kind='stable').astype(np.int32)
        return order, np.searchsorted(ids[order], np.arange(nids + 2)).astype(np.int64)

    def cooccurrence(self, packed):
        mine = self.pts[self.lo:self.hi]
        m = np.stack([O.points_in_obb(mine, b[0:3], b[3:12].reshape(3, 3), b[12:15]) for b in packed]).astype(np.float32)
        return ((m @ m.T) > 0).astype(np.uint8)


def test_merge_bb_partner_search_on_arrays_keeps_the_reference_control_flow():
    """merge_bb's partner loop runs on arrays that mirror info_sem (parent of the CURRENT list entry, member counts, box bounds) while
    entries are deleted and ids relabelled.  Randomised scenes with what the literal loop of merge_intersecting_bb.py:68-91,103-137 is
    sensitive to -- instances with fewer than 4 points (the early return), empty instances, ids beyond the info list, parents of any
    hashable type, heavy overlap (many deletions shifting the list under the running index) -- against the oracle's literal restatement."""
    import copy
    from Fusion3DSeg.merge_intersecting_bb import merge_bb
    merged = 0
    for seed in range(24):
        rng = np.random.default_rng(seed)
        nb = int(rng.integers(3, 40))
        centres = rng.uniform(0, rng.uniform(1.0, 6.0), (nb, 3))
        sizes = rng.choice([0, 1, 3, 4, 5, 40, 120], nb, p=[0.05, 0.05, 0.1, 0.1, 0.1, 0.3, 0.3])
        pts = np.vstack([centres[k] + rng.normal(size=(int(sizes[k]), 3)) * rng.uniform(0.05, 0.8, 3) for k in range(nb)] + [np.zeros((0, 3))])
        ids = np.repeat(np.arange(nb), sizes).astype(np.int64)
        if len(pts) < 8:
            continue
        perm = rng.permutation(len(pts))
        pts, ids = pts[perm], ids[perm]
        nparents = int(rng.integers(1, 4))
        kind = int(rng.integers(3))
        parent = [(k % nparents) if kind == 0 else str(k % nparents) if kind == 1 else (k % nparents, 'p') for k in range(nb)]
        if kind == 2:
            parent = [p[0] * 1.5 for p in parent]                       # floats
        ninfo = nb if rng.random() < 0.7 else max(2, nb - int(rng.integers(1, 3)))       # ids beyond the info list
        info = [{'id': k, 'category_id': 86, 'parent_id': parent[k], 'area': int((ids == k).sum())} for k in range(ninfo)]
        want_info, want_ids = O.merge_bb(copy.deepcopy(info), ids.copy(), pts)
        got_info, got_ids = merge_bb(None, copy.deepcopy(info), ids.copy(), pts, box_fn=O.obb_from_points, backend=_NumpyCloud)
        merged += len(info) - len(want_info)
        assert np.array_equal(got_ids, want_ids), seed
        assert [(d['id'], d['area']) for d in got_info] == [(d['id'], d['area']) for d in want_info], seed
        for g, w in zip(got_info, want_info):
            assert ('bbox' in g) == ('bbox' in w), seed
            if 'bbox' in w:
                assert np.array_equal(np.array(g['bbox']), np.array(w['bbox'])), seed
    assert merged > 40                                                 # the scenes really merge (and delete) a lot
