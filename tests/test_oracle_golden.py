"""The NumPy oracle (oracle/np_ref.py) against vectors produced by running the reference
(tests/golden/make_golden.py) and against SURVEY 8(a)'s known answers.  CPU only."""
import json

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import np_ref as O

UV_MAX_FLIP_FRACTION = 1e-6        # stated tolerance: |du|,|dv| <= 1 px on <= 1e-6 of samples


def test_rotate_matches_reference(golden):
    g = golden('rotate')
    for q, want in zip(g['q_wxyz'], g['rotated']):
        got = O.rotate(q, g['points'])
        scale = np.dot(q, q) * np.abs(g['points']).max()
        # the reference's np.dot goes through BLAS (order/FMA CPU-dependent): a few ulp
        assert np.abs(got - want).max() <= 8 * np.finfo(float).eps * scale


def test_points2pixel_matches_reference(golden):
    g = golden('points2pixel')
    n_total = n_flip = 0
    for ki, K in enumerate(g['K']):
        for j, (q, t) in enumerate(zip(g['q_wxyz'], g['t'])):
            got = O.points2pixel(g['points'], K, q, t)
            want = g['uv'][ki, j]
            defined = np.abs(want.astype(np.int64)) < 2 ** 30          # int32 cast of huge values is UB in C
            d = np.abs(got.astype(np.int64) - want)[:, defined[0] & defined[1]]
            assert d.max() <= 1
            n_total += d.size
            n_flip += int((d != 0).sum())
    assert n_flip <= max(0, int(UV_MAX_FLIP_FRACTION * n_total)), (n_flip, n_total)


def test_known_answers_from_survey():
    ka = json.loads((GOLDEN / 'survey_known_answers.json').read_text())
    p = ka['points2pixel']
    K = np.array(p['K'])
    for scale in (1.0, 2.0):
        uv = O.points2pixel(np.array(p['points'], float), K, scale * np.array(p['q_wxyz']), p['t'])
        assert uv.tolist() == p['uv']
    eyes, look, so, fn = O.frustum_data(K, ka['frustum']['w'], ka['frustum']['h'], [p['q_wxyz']], [p['t']])
    assert np.allclose(look[0], ka['frustum']['lookat'], rtol=0, atol=1e-15)
    ppts, pnrm = O.frustum_planes(K, 720, 960, [p['q_wxyz']], [p['t']], ka['inside']['max_depth'])
    assert O.point_inside_polyhedra(np.array(p['points'], float), ppts[0], pnrm[0]).tolist() == ka['inside']['inside']
    s = ka['segment']
    for case in s['cases']:
        got = O.segment(np.array(s['votes'], float), s['nclasses'], case['threshold'], case['filter'])
        assert got.tolist() == case['classes']


def test_frustum_data_matches_reference(golden):
    g = golden('frustum')
    for name in ('calib', 'sq512', 'skew'):
        w, h = g[f'{name}_wh']
        e, l, so, fn = O.frustum_data(g[f'{name}_K'], int(w), int(h), g['q_wxyz'], g['t'])
        for got, key in ((e, 'eyes'), (l, 'lookats'), (so, 'spoke_origins'), (fn, 'face_normals')):
            want = g[f'{name}_{key}']
            assert got.shape == want.shape
            # np.linalg.inv (LAPACK) vs adjugate, BLAS dot in rotate: rounding-level differences only
            assert np.abs(got - want).max() <= 1e-12
    w, h = g['calib_wh']
    e, l, so, fn = O.frustum_data(g['calib_K'], int(w), int(h), g['q_wxyz'], g['t'], g['perm_ids'])
    assert np.abs(e - g['perm_eyes']).max() <= 1e-12
    assert np.abs(so - g['perm_spoke_origins']).max() <= 1e-12      # double lookup eyes[ids][ids]
    assert np.abs(fn - g['perm_face_normals']).max() <= 1e-12


def test_inside_polyhedra_bit_exact(golden):
    g = golden('inside_polyhedra')
    for j in range(len(g['plane_points'])):
        got = O.point_inside_polyhedra(g['points'], g['plane_points'][j], g['plane_normals'][j])
        assert np.array_equal(got, g['inside'][j])
        got = O.point_inside_polyhedra(g['adv_points'], g['plane_points'][j], g['plane_normals'][j])
        assert np.array_equal(got, g['inside_adv'][j])              # points within a few ulp of a plane
    assert g['inside'].any() and not g['inside'].all()
    assert g['inside_adv'].any() and not g['inside_adv'].all()


def test_vote_scatter_q1_and_segment(golden):
    g = golden('voting')
    ncls = int(g['nclasses'])
    votes = np.zeros_like(g['votes'])
    for mask, lut in zip(g['masks'], g['uv2pt']):
        O.vote_frame(votes, lut, mask.reshape(-1))
    assert np.array_equal(votes, g['votes'])
    assert votes[7].max() <= len(g['masks'])                         # Q1: +1 per frame, not per pixel
    for i in range(int(g['nseg'])):
        flt = g[f'seg{i}_filter'].tolist() if g[f'seg{i}_has_filter'] else None
        got = O.segment(g['votes'], ncls, float(g[f'seg{i}_threshold']), flt)
        assert np.array_equal(got, g[f'seg{i}_classes']), i
    got = O.segment(g['votes'], g['votes'].shape[1], 0.75, None)      # Q2
    assert np.array_equal(got, g['segq2_classes'])
    for i in range(int(g['nsmall'])):
        flt = g[f'small{i}_filter'].tolist() if g[f'small{i}_has_filter'] else None
        got = O.segment(g['small_votes'], 4, float(g[f'small{i}_threshold']), flt)
        assert np.array_equal(got, g[f'small{i}_classes']), i


def test_vote_errors_like_numpy():
    votes = np.zeros((4, 3))
    with pytest.raises(IndexError):
        O.vote_frame(votes, np.array([0, 1], np.int32), np.array([0, 3], np.uint8))
    with pytest.raises(IndexError):
        O.vote_frame(votes, np.array([4], np.int32), np.array([0], np.uint8))
    O.vote_frame(votes, np.array([-2, -1], np.int32), np.array([1, 1], np.uint8))    # -2 wraps, -1 is "none"
    assert votes[2, 1] == 1 and votes.sum() == 1


def test_filter_remap_table_equals_sequential_loop():
    for flt in ([2, 0, 1], [86, 114, 115], [1, 1, 0], [3, 3, 3, 0]):
        tab = O.filter_remap_table(133, flt, 134)
        x = np.arange(134)
        y = x.copy()
        for i, c in enumerate(flt):
            y[y == i] = c
        assert np.array_equal(tab[x], y)


# --------------------------------------------------------------------------- C oracle == NumPy oracle, bit for bit
def test_c_oracle_equals_numpy_oracle(golden):
    from oracle import c_ref as Cc
    from f3d import synth
    g = golden('points2pixel')
    pts = g['points']
    for q, t in zip(g['q_wxyz'], g['t']):
        assert np.array_equal(Cc.rotate(q, pts), O.rotate(q, pts))
        for K in g['K']:
            assert np.array_equal(Cc.points2pixel(pts, K, q, t), O.points2pixel(pts, K, q, t))
    pp, pn, eyes, look = Cc.frustum_planes(g['K'][0], 720, 960, g['q_wxyz'], g['t'], 4.0)
    wp, wn = O.frustum_planes(g['K'][0], 720, 960, g['q_wxyz'], g['t'], 4.0)
    assert np.array_equal(pp, wp) and np.array_equal(pn, wn)
    gi = golden('inside_polyhedra')
    for j in range(len(gi['plane_points'])):
        assert np.array_equal(Cc.point_inside_polyhedra(gi['adv_points'], gi['plane_points'][j], gi['plane_normals'][j]), gi['inside_adv'][j])
    gv = golden('voting')
    votes = np.zeros_like(gv['votes'])
    for mask, lut in zip(gv['masks'], gv['uv2pt']):
        Cc.vote_frame(votes, lut, mask.reshape(-1))
    assert np.array_equal(votes, gv['votes'])
    for i in range(int(gv['nseg'])):
        flt = gv[f'seg{i}_filter'].tolist() if gv[f'seg{i}_has_filter'] else None
        assert np.array_equal(Cc.segment(gv['votes'], 133, float(gv[f'seg{i}_threshold']), flt), gv[f'seg{i}_classes'])
    sc = synth.scene('C1', n=4000, mask_kind='iid')
    for thr, flt in [(0.5, None), (0.3, [115, 0, 86])]:
        want, wv = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                         133, thr, flt, return_votes=True)
        got, gvv = Cc.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                          133, thr, flt, return_votes=True)
        assert np.array_equal(got, want) and np.array_equal(gvv, wv)


def test_radius_adjacency_restates_sklearn_kdtree():
    """(f)#1: the oracle's pairwise test against the library the reference calls (sklearn KDTree, fusion.py:374-375),
    including lattices whose neighbour distances equal the radius exactly."""
    from sklearn.neighbors import KDTree
    rng = np.random.default_rng(5)
    lattice = np.stack(np.meshgrid(np.arange(7.), np.arange(6.), np.arange(4.), indexing='ij'), -1).reshape(-1, 3)
    cases = [(rng.uniform(-1, 1, (600, 3)), 0.25), (lattice, 1.0), (lattice, np.sqrt(2.0)), (lattice * 0.1, 0.1),
             (lattice * 0.05, 2 * 0.05), (np.repeat(rng.uniform(0, 1, (40, 3)), 3, axis=0), 0.0)]
    for P, r in cases:
        want = KDTree(P).query_radius(P, r=r)
        got = O.radius_adjacency(P, r)
        assert all(np.array_equal(np.sort(w), g) for w, g in zip(want, got)), r


def test_unproject_depth_matches_reference_getModP3d(golden):
    """(f)#3: oracle vs the reference's RTAB2Cache.__getModP3d (ios_rtab.py:179-193) on camera points produced by the restated
    :171-173; tolerance = the rotate golden's (the reference's np.dot / np.cross go through BLAS)."""
    g = golden('modp3d')
    for d, q, t, orig, want in zip(g['depths'], g['odo_xyzw'], g['odo_xyz'], g['orig_ptx'], g['mod_ptx']):
        assert np.array_equal(O.unproject_depth(d, g['K'], [1.0, 0, 0, 0], np.zeros(3), depth_scale=1), orig)
        got = O.unproject_depth(d, g['K'], q[[3, 0, 1, 2]], t)
        scale = np.dot(q, q) * np.abs(orig / 1000).max() + np.abs(t).max()
        assert np.abs(got - want).max() <= 8 * np.finfo(float).eps * scale


SEM_CASES = ('img', 'edge', 'tie', 'tie0', 'tie5', 'small')
SEM_BAND = 1e-5            # relative half-width of the band around conf_threshold that is not compared (see below)


def sem_compare(got, g, name):
    """`got` against the reference's own post-processing (tests/golden/make_golden_sem.py ran get2DSeg.py:111-120 with
    CPU torch).  Labels must be EQUAL wherever torch's float32 max-probability is farther than SEM_BAND * threshold from
    the threshold: the restatement and the HIP kernel sum the exponentials in another order (and the kernel uses
    v_exp_f32), about 1e-6 relative on the probability, so only a pixel that close to the threshold may differ."""
    want, pmax, thr = g[f'{name}_mask'], g[f'{name}_pmax'], float(g[f'{name}_conf'])
    band = (np.abs(pmax.astype(np.float64) - thr) <= SEM_BAND * thr) if thr else np.zeros(want.shape, bool)
    assert band.mean() < 1e-3, name                                   # the band is (nearly) empty: the comparison is not vacuous
    assert np.array_equal(np.asarray(got, np.int64)[~band], want[~band]), name
    return band


def test_sem_logits_to_mask_matches_torch_reference(golden):
    g = golden('sem_mask')
    for name in SEM_CASES:
        thr = float(g[f'{name}_conf'])
        sem_compare(O.sem_logits_to_mask(g[f'{name}_sem'], thr, 133), g, name)
    # the fixture really exercises the cases it is named for
    assert (g['tie0_mask'][0] == 7).all() and (g['tie0_mask'][1] == 0).all() and (g['tie0_mask'][2] == 0).all()
    assert (g['tie_mask'][1] == 133).all() and (g['tie5_mask'][1] == 1).all()
    rel = np.abs(g['edge_pmax'].astype(np.float64) - 0.017) / 0.017
    assert rel.min() > 2 * SEM_BAND and (rel < 2e-4).mean() > 0.3     # straddlers just outside the band, on both sides
    assert 0.3 < (g['edge_mask'] == 133).mean() < 0.7
