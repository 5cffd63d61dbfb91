#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference here.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):      python tests/golden/make_golden.py

What is executed, and how much of it is the reference's own arithmetic
----------------------------------------------------------------------
* ``Fusion3DSeg/intersections.py`` and ``Fusion3DSeg/segUtils/cv.py`` import
  as they are (numpy + einops only) and are called directly.
* ``RTAB_utils/spatQuad.py``, ``Fusion3DSeg/camera_utils.py``,
  ``Fusion3DSeg/fusion.py`` and ``Fusion3DSeg/segUtils/voting.py`` cannot be
  imported: their module headers import third-party packages that are absent
  from this image (pyquaternion, cv2, open3d).  Nothing is installed or
  stubbed into ``sys.modules``.  Instead the *function / class definitions*
  this path needs are compiled straight from the reference's files at
  generation time (``_defs_from``) and run unmodified in a namespace holding
  only numpy/einops.  None of the functions captured here calls cv2 or
  open3d.
* pyquaternion (unpinned by the reference, absent here) contributes exactly
  three members at the reference's call sites (camera_utils.py:22,128):
  ``Quaternion(seq4)``, ``.elements`` and ``.inverse``.  ``_QuaternionBase``
  below restates those three from pyquaternion's published definition
  (elements = the four numbers as float64, NOT normalised;
  inverse = conjugate / sum of squares).  That restated piece is
  "parity unpinned"; everything downstream of it (rotate, subtract, K@,
  divide, floor, astype) is the reference's code.  ``rotate`` itself is
  additionally captured with ``elements`` supplied as plain data, which
  involves no restated code at all.
* SURVEY.md section 8(a) lists known answers the survey stage captured from
  the reference; they are committed verbatim in ``survey_known_answers.json``
  and this script asserts that what it produces agrees with them.

Outputs are plain ``.npz`` / ``.json`` (no pickles).
"""
import ast
import json
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
from einops import rearrange, repeat

REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))


def _defs_from(relpath, names, namespace):
    """Compile the named top-level defs/classes of a reference file into `namespace`."""
    src = (REF / relpath).read_text()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(keep) == len(names), (relpath, names)
    mod = ast.Module(body=keep, type_ignores=[])
    exec(compile(mod, str(REF / relpath), 'exec'), namespace)
    return namespace


class _QuaternionBase:
    """The three pyquaternion members the path uses (restated; see module docstring)."""

    def __init__(self, seq):
        self.q = np.array(seq, dtype=np.float64)
        assert self.q.shape == (4,)

    @property
    def elements(self):
        return self.q

    @property
    def inverse(self):
        ss = np.dot(self.q, self.q)
        if ss == 0:
            raise ZeroDivisionError("a zero quaternion cannot be inverted")
        conj = np.hstack((self.q[0], -self.q[1:4]))
        return self.__class__(conj / ss)


def load_reference():
    from Fusion3DSeg import intersections as ref_isect          # imports as-is
    from Fusion3DSeg.segUtils import cv as ref_cv                 # imports as-is

    ns_q = {'np': np, 'Quaternion': _QuaternionBase}
    _defs_from('RTAB_utils/spatQuad.py', ['SpatQuadranion'], ns_q)
    SpatQ = ns_q['SpatQuadranion']

    ns_cam = {'np': np, 'repeat': repeat, 'rearrange': rearrange, 'SpatQuadranion': SpatQ}
    _defs_from('Fusion3DSeg/camera_utils.py',
               ['points2pixel', 'pixel2point', 'get_camera_frustum', 'camera2world',
                'get_frustum_unit_vectors', 'get_frustum_face_normals'], ns_cam)
    cam = SimpleNamespace(**{k: v for k, v in ns_cam.items() if callable(v)})

    ns_fus = {'np': np, 'repeat': repeat, 'rearrange': rearrange, 'cam_utils': cam,
              'point_inside_polyhedra': ref_isect.point_inside_polyhedra, 'Path': Path}
    _defs_from('Fusion3DSeg/fusion.py', ['Fusion'], ns_fus)

    ns_vote = {'np': np, 'Path': Path}
    _defs_from('Fusion3DSeg/segUtils/voting.py', ['VotingSegmentation'], ns_vote)
    return ref_isect, ref_cv, SpatQ, cam, ns_fus['Fusion'], ns_vote['VotingSegmentation']


# ----------------------------------------------------------------------------
# shared synthetic scene (same recipe as SURVEY 8(d), small)
# ----------------------------------------------------------------------------
CALIB_K = np.array([[798.94403076171875, 0., 361.95578002929688],
                    [0., 798.94403076171875, 474.56329345703125],
                    [0., 0., 1.]])


def look_at_quat(eye, target):
    """camera->world unit quaternion (w,x,y,z): camera x right, y down, z forward."""
    z = target - eye
    z = z / np.linalg.norm(z)
    up = np.array([0., 0., 1.])
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def ring_views(V, rng):
    th = 2 * np.pi * np.arange(V) / V
    eyes = np.stack([4 * np.cos(th), 4 * np.sin(th), np.full(V, 1.5)], axis=1)
    targets = np.array([0., 0., 1.5]) + rng.uniform(-0.5, 0.5, (V, 3))
    quats = np.stack([look_at_quat(e, t) for e, t in zip(eyes, targets)])
    return quats, eyes


def main():
    ref_isect, ref_cv, SpatQ, cam, Fusion, Voting = load_reference()
    rng = np.random.default_rng(20240611)
    known = json.loads((OUT / 'survey_known_answers.json').read_text())

    # ------------------------------------------------------------------ rotate
    q_list = np.array([[0.5, 0.5, -0.5, 0.5],
                       [1.0, 0.0, 0.0, 0.0],
                       [0.3, -1.2, 0.7, 2.0],       # not unit: rotate scales by |q|^2
                       [-0.18257418583505536, 0.3651483716701107, 0.5477225575051661, 0.7302967433402214]])
    p_rot = np.vstack([rng.uniform(-5, 5, (500, 3)),
                       (rng.uniform(-5, 5, (12, 3))).astype(np.float32).astype(np.float64),
                       np.zeros((1, 3))])
    rot_out = np.stack([SpatQ.rotate(SimpleNamespace(elements=q), p_rot) for q in q_list])
    np.savez_compressed(OUT / 'rotate.npz', q_wxyz=q_list, points=p_rot, rotated=rot_out)

    # ------------------------------------------------------------- points2pixel
    V = 6
    quats, eyes = ring_views(V, rng)
    quats[1] *= 2.0                                   # Q4: un-normalised pose
    quats[2] *= -0.37
    pts = rng.uniform([-5, -5, 0], [5, 5, 3], (2500, 3)).astype(np.float32).astype(np.float64)
    pts64 = rng.uniform([-5, -5, 0], [5, 5, 3], (500, 3))          # full f64 mantissas
    pts = np.vstack([pts, pts64])
    K_sq = np.array([[400., 0., 256.], [0., 400., 256.], [0., 0., 1.]])
    K_skew = np.array([[812.5, 1.75, 500.25], [0.125, 790.0, 515.5], [0., 0., 1.]])
    Ks = np.stack([CALIB_K, K_sq, K_skew])
    uv = np.zeros((len(Ks), V, 2, len(pts)), np.int32)
    zc = np.zeros((len(Ks), V, len(pts)))
    with np.errstate(all='ignore'):
        for ki, K in enumerate(Ks):
            for j in range(V):
                uv[ki, j] = cam.points2pixel(pts, K, quats[j], eyes[j])
                # camera-space depth, used by the tests only to skip the
                # |u|,|v| >= 2^31 entries whose int32 cast is undefined
                zc[ki, j] = SpatQ(quats[j]).inverse.rotate(pts - eyes[j])[:, 2]
    np.savez_compressed(OUT / 'points2pixel.npz', points=pts, K=Ks, q_wxyz=quats, t=eyes, uv=uv, zcam=zc.astype(np.float32))

    ka = known['points2pixel']
    got = cam.points2pixel(np.array(ka['points']), CALIB_K, np.array(ka['q_wxyz']), np.array(ka['t']))
    assert got.tolist() == ka['uv'], (got.tolist(), ka['uv'])
    got2 = cam.points2pixel(np.array(ka['points']), CALIB_K, 2 * np.array(ka['q_wxyz']), np.array(ka['t']))
    assert got2.tolist() == ka['uv']

    # ------------------------------------------------------------ frustum data
    fr = {}
    for name, K, w, h in [('calib', CALIB_K, 720, 960), ('sq512', K_sq, 512, 512), ('skew', K_skew, 1024, 1024)]:
        e, l, so, fn = Fusion._get_frustum_data(K, w, h, quats, eyes)
        fr[f'{name}_K'] = K
        fr[f'{name}_wh'] = np.array([w, h])
        fr[f'{name}_eyes'], fr[f'{name}_lookats'] = e, l
        fr[f'{name}_spoke_origins'], fr[f'{name}_face_normals'] = so, fn
    # frame_ids given: fusion.py:127-129 indexes eyes twice (spoke origins = eyes[ids][ids]);
    # a strict subset such as [4, 0, 2] raises IndexError there, a permutation shows the double lookup
    perm = np.array([4, 0, 2, 5, 1, 3])
    e, l, so, fn = Fusion._get_frustum_data(CALIB_K, 720, 960, quats, eyes, perm)
    fr['perm_ids'] = perm
    fr['perm_eyes'], fr['perm_lookats'], fr['perm_spoke_origins'], fr['perm_face_normals'] = e, l, so, fn
    fr['q_wxyz'], fr['t'] = quats, eyes
    np.savez_compressed(OUT / 'frustum.npz', **fr)
    e1, l1, _, _ = Fusion._get_frustum_data(CALIB_K, 720, 960, np.array([ka['q_wxyz']]), np.array([ka['t']]))
    assert np.allclose(l1[0], known['frustum']['lookat'], rtol=0, atol=1e-15), l1

    # ------------------------------------- point_inside_polyhedra (fusion.py:254-260)
    max_depth = 4.0
    poly = {'points': pts, 'max_depth': np.array(max_depth)}
    e, l, so, fn = Fusion._get_frustum_data(CALIB_K, 720, 960, quats, eyes)
    ppts = np.concatenate([so, (e + max_depth * l)[:, None, :]], axis=1)       # [V,5,3]
    pnrm = np.concatenate([fn, (-l)[:, None, :]], axis=1)
    inside = np.stack([ref_isect.point_inside_polyhedra(pts, ppts[j], pnrm[j]) for j in range(V)])
    # adversarial: points within a few ulp of each plane, where the summation
    # order of the 3-term dot product decides the sign
    adv = []
    for j in range(V):
        for m in range(5):
            n = pnrm[j, m]
            t1 = np.cross(n, [0.3, -0.2, 0.9]); t1 /= np.linalg.norm(t1)
            t2 = np.cross(n, t1)
            ab = rng.uniform(-3, 3, (60, 2))
            base = ppts[j, m] + ab[:, :1] * t1 + ab[:, 1:] * t2
            k = rng.integers(-3, 4, (60, 1)) * 2.0 ** -51
            adv.append(base + k * n)
    adv = np.vstack(adv)
    inside_adv = np.stack([ref_isect.point_inside_polyhedra(adv, ppts[j], pnrm[j]) for j in range(V)])
    poly.update(plane_points=ppts, plane_normals=pnrm, inside=inside, adv_points=adv, inside_adv=inside_adv)
    np.savez_compressed(OUT / 'inside_polyhedra.npz', **poly)
    ki = known['inside']
    e, l, so, fn = Fusion._get_frustum_data(CALIB_K, 720, 960, np.array([ka['q_wxyz']]), np.array([ka['t']]))
    got = ref_isect.point_inside_polyhedra(
        np.array(ka['points']), np.vstack([so[0], e[0] + ki['max_depth'] * l[0]]), np.vstack([fn[0], -l[0]]))
    assert got.tolist() == ki['inside']

    # ---------------------------------------------------------------- voting
    class InjectedVoting(Voting):
        """Reference vote()/segment() with frames handed over as arrays instead of files."""
        def __init__(self, npts, depth_hw, nclasses, frames):
            self.npts, self.depth_hw, self.nclasses = npts, depth_hw, nclasses
            self.votes = np.zeros((npts, nclasses + 1))
            self.frames = frames
            self.nframes = len(frames)

        def _read_data(self, idx):
            return self.frames[idx]

    vt = {}
    npts, h, w, ncls = 300, 24, 32, 133
    frames = []
    for f in range(5):
        mask = rng.integers(0, 134, (h, w)).astype(np.uint8)
        if f % 2:                                         # blocky masks -> real pluralities
            mask = np.repeat(np.repeat(rng.choice([0, 15, 86, 114, 115, 120, 132, 133], (h // 8, w // 8)), 8, 0), 8, 1).astype(np.uint8)
        lut = rng.integers(-1, npts, h * w).astype(np.int32)
        lut[rng.random(h * w) < 0.3] = -1
        lut[:40] = 7                                       # Q1: many pixels -> same point
        frames.append((mask, lut))
    frames.append((rng.integers(0, 134, (h, w)).astype(np.uint8), np.full(h * w, -1, np.int32)))   # all invalid
    voter = InjectedVoting(npts, (h, w), ncls, frames)
    votes = voter.vote(resize=False).copy()
    vt['masks'] = np.stack([m for m, _ in frames])
    vt['uv2pt'] = np.stack([u for _, u in frames])
    vt['votes'] = votes
    vt['nclasses'] = np.array(ncls)
    seg_cases = [(0.5, None), (0.0, None), (0.5, [86, 114, 115]), (0.3, [115, 0, 86]), (0.75, None),
                 (0.5, [2, 0, 1]), (0.2, [1, 1, 0]), (0.34, [133, 15])]
    for i, (thr, flt) in enumerate(seg_cases):
        vt[f'seg{i}_threshold'] = np.array(thr)
        vt[f'seg{i}_filter'] = np.array([] if flt is None else flt, np.int64)
        vt[f'seg{i}_has_filter'] = np.array(flt is not None)
        vt[f'seg{i}_classes'] = voter.segment(thr, flt)
    vt['nseg'] = np.array(len(seg_cases))
    # Q2: a voter re-created from a votes file has nclasses = votes.shape[1]
    reloaded = Voting.__new__(Voting)
    reloaded.votes = votes.copy()
    reloaded.nclasses = votes.shape[1]
    vt['segq2_classes'] = reloaded.segment(0.75, None)
    # hand-made small-alphabet table, incl. ties (first max wins) and Q3 aliasing
    small = np.array([[0, 0, 0, 0, 0], [2, 2, 0, 0, 0], [0, 1, 3, 0, 0], [1, 0, 0, 0, 3], [0, 0, 1, 1, 2],
                      [5, 5, 5, 5, 5], [0, 0, 0, 0, 7], [1, 0, 0, 0, 0], [0, 3, 0, 3, 0]], np.float64)
    sv = Voting.__new__(Voting)
    sv.votes, sv.nclasses = small, 4
    small_cases = [(0.5, None), (0.5, [2, 0, 1]), (0.3, [2, 3]), (0.0, None), (0.2, [3, 1]), (0.5, [4, 0]), (1.0, None)]
    vt['small_votes'] = small
    for i, (thr, flt) in enumerate(small_cases):
        vt[f'small{i}_threshold'] = np.array(thr)
        vt[f'small{i}_filter'] = np.array([] if flt is None else flt, np.int64)
        vt[f'small{i}_has_filter'] = np.array(flt is not None)
        vt[f'small{i}_classes'] = sv.segment(thr, flt)
    vt['nsmall'] = np.array(len(small_cases))
    np.savez_compressed(OUT / 'voting.npz', **vt)
    for case in known['segment']['cases']:
        sv.votes, sv.nclasses = np.array(known['segment']['votes'], np.float64), known['segment']['nclasses']
        assert sv.segment(case['threshold'], case['filter']).tolist() == case['classes']

    # -------------------------------------------------- intersections primitives
    it = {}
    o, d = rng.normal(size=3), rng.normal(size=3)
    st, en = rng.normal(size=(50, 3)), rng.normal(size=(50, 3))
    it['rxl_origin'], it['rxl_direction'], it['rxl_starts'], it['rxl_ends'] = o, d, st, en
    it['rxl_points'], it['rxl_within'] = ref_isect.ray_x_lines(o, d, st, en)
    pp, pn = rng.normal(size=3), rng.normal(size=3); pn /= np.linalg.norm(pn)
    og, dr = rng.normal(size=(50, 3)), rng.normal(size=(50, 3)); dr /= np.linalg.norm(dr, axis=1)[:, None]
    it['rxp_plane_point'], it['rxp_plane_normal'], it['rxp_origins'], it['rxp_directions'] = pp, pn, og, dr
    it['rxp_points'], it['rxp_valid'] = ref_isect.rays_x_plane(pp, pn, og, dr)
    # intersections.py:89-90 subtracts [N,3] from [N,M,3] without a new axis: it raises for
    # N != M (recorded below) and pairs line n with plane index n when N == M.  Dead code (8(a) a12).
    lo, le = rng.normal(size=(6, 3)) * 2, rng.normal(size=(6, 3)) * 2
    pps, pns = rng.normal(size=(6, 3)), rng.normal(size=(6, 3)); pns /= np.linalg.norm(pns, axis=1)[:, None]
    it['lxp_origins'], it['lxp_ends'], it['lxp_plane_points'], it['lxp_plane_normals'] = lo, le, pps, pns
    it['lxp_points'], it['lxp_valid'] = ref_isect.lines_x_planes(lo, le, pps, pns)
    try:
        ref_isect.lines_x_planes(np.vstack([lo, lo]), np.vstack([le, le]), pps, pns)
        it['lxp_n_ne_m_error'] = np.array('')
    except Exception as exc:
        it['lxp_n_ne_m_error'] = np.array(type(exc).__name__)
    verts = np.array([[0, 0, 0], [2, 0, 0], [2, 1, 0], [0, 1, 0]], np.float64)
    pg = np.c_[rng.uniform(-1, 3, (200, 2)), np.zeros(200)]
    it['pip_points'], it['pip_vertices'] = pg, verts
    it['pip_inside'], it['pip_within'] = ref_isect.point_inside_polygon(pg, verts)
    n1, n2, la = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
    v1, v2 = rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
    it['pxp_n1'], it['pxp_n2'], it['pxp_lookat'], it['pxp_v1'], it['pxp_v2'] = n1, n2, la, v1, v2
    it['pxp_dir_normals'] = ref_isect.plane_x_plane(n1=n1, n2=n2, lookat=la)
    it['pxp_dir_vertices'] = ref_isect.plane_x_plane(v1=v1, v2=v2)
    q = rng.normal(size=(30, 3))
    it['ppp_points'], it['ppp_plane_point'], it['ppp_normal'] = q, pp, pn
    it['ppp_out'] = np.asarray(ref_isect.points_plane_projection(q, pp, pn))
    ls_, le_ = rng.normal(size=(30, 3)), rng.normal(size=(30, 3))
    it['lpp_starts'], it['lpp_ends'] = ls_, le_
    out = ref_isect.lines_plane_projection(ls_, le_, pp, pn)
    for i, a in enumerate(out if isinstance(out, tuple) else (out,)):
        it[f'lpp_out{i}'] = np.asarray(a)
    o1, d1, o2, d2 = (rng.normal(size=(25, 3)) for _ in range(4))
    it['rrc_o1'], it['rrc_d1'], it['rrc_o2'], it['rrc_d2'] = o1, d1, o2, d2
    d2[:3] = d1[:3] + (o2[:3] - o1[:3])                       # parallel segments -> denom == 0
    with np.errstate(all='ignore'):
        outs = [ref_isect.ray_ray_closest(a0, a1, b0, b1) for a0, a1, b0, b1 in zip(o1, d1, o2, d2)]
    it['rrc_pa'] = np.stack([o[0] for o in outs])
    it['rrc_pb'] = np.stack([o[1] for o in outs])
    it['rrc_distance'] = np.array([o[2] for o in outs])
    it['rrc_flags'] = np.array([[bool(o[3]), bool(o[4]), bool(o[5])] for o in outs])
    np.savez_compressed(OUT / 'intersections.npz', **it)

    # ----------------------------------------------------- split_into_instances
    sp = {}
    n = 400
    xy = rng.uniform(0, 10, (n, 2))
    d2 = ((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1)
    adj = [np.nonzero(d2[i] < 0.7 ** 2)[0] for i in range(n)]          # includes self, like query_radius
    cls = rng.choice([86, 114, 115, 133, 3], n, p=[0.3, 0.25, 0.2, 0.15, 0.1]).astype(np.int64)
    flat = np.concatenate(adj)
    offs = np.concatenate([[0], np.cumsum([len(a) for a in adj])])
    sp['classes'], sp['adj_flat'], sp['adj_offsets'] = cls, flat.astype(np.int64), offs.astype(np.int64)
    cases = [(None, 1), ([86, 114, 115], 5), ([86, 114, 115], 100), (None, 8), ([3], 2)]
    for i, (ic, mp) in enumerate(cases):
        insts, ids, info, newcls = ref_cv.split_into_instances(cls, adj, 133, ic, mp)
        sp[f'case{i}_instance_classes'] = np.array([] if ic is None else ic, np.int64)
        sp[f'case{i}_has_instance_classes'] = np.array(ic is not None)
        sp[f'case{i}_minimum_points'] = np.array(mp)
        sp[f'case{i}_ninst'] = np.array(len(insts))
        sp[f'case{i}_ids'] = ids
        sp[f'case{i}_classes'] = newcls
        sp[f'case{i}_info'] = np.array([[d['id'], int(d['isthing']), d['category_id'], d['area']] for d in info], np.int64).reshape(-1, 4)
    sp['ncases'] = np.array(len(cases))
    np.savez_compressed(OUT / 'split_instances.npz', **sp)

    for f in sorted(OUT.glob('*.npz')):
        print(f'{f.name:28s} {f.stat().st_size/1024:8.1f} KiB')


if __name__ == '__main__':
    main()
