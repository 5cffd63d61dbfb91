"""NumPy restatement of the Fusion3DSeg hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does (it fails loudly
when the HIP library is missing).

Every function cites the reference file:line it follows (paths relative to the
reference repository).  Pinning: ``tests/test_oracle_golden.py`` checks each
function against ``tests/golden/*.npz`` (produced by running the reference in
the build container, see ``tests/golden/make_golden.py``) and against the
known answers of SURVEY.md 8(a).  Parts that rest on absent third-party
packages are marked "parity unpinned" where they are defined.

Canonical arithmetic
--------------------
The reference mixes NumPy ufuncs (one IEEE operation per element, never fused)
with BLAS calls (``np.dot`` in rotate, ``@`` in points2pixel) and ``einsum``
whose summation order / FMA use depends on the CPU and the BLAS build, so the
reference itself is reproducible only to the last bits across machines.  The
oracle fixes ONE order, which the C oracle and the HIP kernels follow exactly:

* every product and sum is rounded separately (no FMA contraction);
* 3-term dot products are summed left to right, ``(a0*b0 + a1*b1) + a2*b2``,
  EXCEPT the plane test of ``point_inside_polyhedra`` which uses
  ``(d0*n0 + d2*n2) + d1*n1`` -- the order ``np.einsum('nmc,mc->mn')`` is
  observed to use in the build container (AVX-512 tree reduction), so that the
  committed near-plane golden points match bit for bit;
* IEEE division, ``floor``; float64 -> int32 of NaN / out-of-range values gives
  INT32_MIN (what the reference's C cast yields on x86-64).
"""
import numpy as np

INT32_MIN = np.int32(-2 ** 31)


# ----------------------------------------------------------------------------
# a1  quaternion pieces
# ----------------------------------------------------------------------------
def quat_inverse(q_wxyz):
    """pyquaternion ``Quaternion(seq4).inverse.elements`` as used at camera_utils.py:22.

    conj(q) / (w^2 + x^2 + y^2 + z^2); the input is NOT normalised (quirk Q4).
    pyquaternion is absent from the image: restated from its published
    definition, parity unpinned beyond the SURVEY 8(a) known answers.
    """
    q = np.asarray(q_wxyz, np.float64)
    ss = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]
    if ss == 0:
        raise ZeroDivisionError("a zero quaternion cannot be inverted")
    return np.array([q[0], -q[1], -q[2], -q[3]]) / ss


def _dot3(a0, a1, a2, b0, b1, b2):
    return (a0 * b0 + a1 * b1) + a2 * b2


def _cross(a0, a1, a2, b0, b1, b2):
    # np.cross: cp0 = a1*b2 - a2*b1 ; cp1 = a2*b0 - a0*b2 ; cp2 = a0*b1 - a1*b0
    return a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0


def rotate(q_wxyz, p):
    """SpatQuadranion.rotate (RTAB_utils/spatQuad.py:7-28): q * p * conj(q), un-normalised."""
    q = np.asarray(q_wxyz, np.float64)
    p = np.asarray(p, np.float64)
    rq, v0, v1, v2 = q[0], q[1], q[2], q[3]
    m0, m1, m2 = -v0, -v1, -v2                                    # vq_  (:18)
    p0, p1, p2 = p[:, 0], p[:, 1], p[:, 2]
    rqp = -_dot3(p0, p1, p2, v0, v1, v2)                          # :22
    c0, c1, c2 = _cross(v0, v1, v2, p0, p1, p2)
    a0, a1, a2 = rq * p0 + c0, rq * p1 + c1, rq * p2 + c2         # vqp  (:23)
    d0, d1, d2 = _cross(a0, a1, a2, m0, m1, m2)
    o0 = (rqp * m0 + rq * a0) + d0                                # :27
    o1 = (rqp * m1 + rq * a1) + d1
    o2 = (rqp * m2 + rq * a2) + d2
    return np.stack([o0, o1, o2], axis=1)


# ----------------------------------------------------------------------------
# a2  points2pixel
# ----------------------------------------------------------------------------
def _to_int32(x):
    with np.errstate(invalid='ignore'):
        ok = (x >= -2147483648.0) & (x <= 2147483647.0)            # NaN -> False
        out = np.where(ok, x, 0.0).astype(np.int32)
    out[~ok] = INT32_MIN
    return out


def project_uvz(points, intrinsic, quat, translation):
    """Float stage of points2pixel (camera_utils.py:21-24): returns (u, v, z_cam) float64."""
    P = np.asarray(points, np.float64)
    K = np.asarray(intrinsic, np.float64)
    t = np.asarray(translation, np.float64)
    d = P - t[None, :]                                             # :21
    c = rotate(quat_inverse(quat), d)                              # :22
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    with np.errstate(all='ignore'):
        h0 = (K[0, 0] * x + K[0, 1] * y) + K[0, 2] * z             # :23
        h1 = (K[1, 0] * x + K[1, 1] * y) + K[1, 2] * z
        h2 = (K[2, 0] * x + K[2, 1] * y) + K[2, 2] * z
        return h0 / h2, h1 / h2, h2                                # :24


def points2pixel(points, intrinsic, quat, translation):
    """camera_utils.py:9-26 -> int32 [2, N] (row 0 = u = column, row 1 = v = row).

    No z > 0 test and no bounds test (quirk Q5).
    """
    u, v, _ = project_uvz(points, intrinsic, quat, translation)
    with np.errstate(all='ignore'):
        return np.stack([_to_int32(np.floor(u)), _to_int32(np.floor(v))])   # :25


# ----------------------------------------------------------------------------
# a3  per-view frustum planes (host side, tiny)
# ----------------------------------------------------------------------------
def inv3(K):
    """3x3 inverse by the adjugate (the reference uses np.linalg.inv = LAPACK, camera_utils.py:75;
    the two agree to rounding, so frustum goldens are compared with a tolerance)."""
    K = np.asarray(K, np.float64)
    a, b, c = K[0]; d, e, f = K[1]; g, h, i = K[2]
    A, B, C = e * i - f * h, c * h - b * i, b * f - c * e
    D, E, F = f * g - d * i, a * i - c * g, c * d - a * f
    G, H, I = d * h - e * g, b * g - a * h, a * e - b * d
    det = (a * A + b * D) + c * G
    return np.array([[A, B, C], [D, E, F], [G, H, I]]) / det


def frustum_data(K, w, h, wxyzs, translations, frame_ids=None):
    """Fusion._get_frustum_data (fusion.py:119-132) over get_camera_frustum (camera_utils.py:60-93),
    camera2world(rescale=1) (:96-132), get_frustum_unit_vectors (:135-150) and
    get_frustum_face_normals (:153-171).

    Returns eyes [F,3], lookats [F,3], spoke_origins [F,4,3], face_normals [F,4,3].
    With ``frame_ids`` the reference indexes the eyes twice for the spoke
    origins (fusion.py:127-129); reproduced.
    """
    wxyzs = np.atleast_2d(np.asarray(wxyzs, np.float64))
    ts = np.atleast_2d(np.asarray(translations, np.float64))
    Kinv = inv3(K)
    pix = np.array([[0, 0, 0], [0, 0, 1], [w, 0, 1], [w, h, 1], [0, h, 1], [w / 2, h / 2, 1]], np.float64)
    cam = np.stack([(Kinv[0, 0] * pix[:, 0] + Kinv[0, 1] * pix[:, 1]) + Kinv[0, 2] * pix[:, 2],
                    (Kinv[1, 0] * pix[:, 0] + Kinv[1, 1] * pix[:, 1]) + Kinv[1, 2] * pix[:, 2],
                    (Kinv[2, 0] * pix[:, 0] + Kinv[2, 1] * pix[:, 1]) + Kinv[2, 2] * pix[:, 2]], axis=1)
    cam = cam / 1                                                    # rescale=1 (:111)
    world = np.stack([rotate(q, cam) + t[None, :] for q, t in zip(wxyzs, ts)])     # [F,6,3] (:127-130)
    eyes = world[:, 0, :]
    vecs = world[:, 1:, :] - world[:, 0:1, :]
    norm = np.sqrt((vecs[..., 0] * vecs[..., 0] + vecs[..., 1] * vecs[..., 1]) + vecs[..., 2] * vecs[..., 2])
    dirs = vecs / norm[..., None]
    lookats = dirs[:, -1, :]
    corners = world[:, 1:-1, :]                                       # [F,4,3]
    a = corners - eyes[:, None, :]
    b = np.roll(corners, -1, axis=1) - eyes[:, None, :]
    n0, n1, n2 = _cross(a[..., 0], a[..., 1], a[..., 2], b[..., 0], b[..., 1], b[..., 2])
    nn = np.sqrt((n0 * n0 + n1 * n1) + n2 * n2)
    normals = np.stack([n0 / nn, n1 / nn, n2 / nn], axis=-1)
    if frame_ids is None:
        frame_ids = np.arange(len(ts))
    eyes = eyes[frame_ids]
    lookats = lookats[frame_ids]
    spoke = np.repeat(eyes[frame_ids][:, None, :], 4, axis=1)
    return eyes, lookats, spoke, normals[frame_ids]


def frustum_planes(K, w, h, wxyzs, translations, max_depth):
    """The 5 planes per view as Fusion.fuse builds them (fusion.py:254-258):
    4 side planes through the eye + the far plane at eye + max_depth*lookat, normal -lookat."""
    eyes, lookats, spoke, normals = frustum_data(K, w, h, wxyzs, translations)
    far_pt = eyes + max_depth * lookats
    pts = np.concatenate([spoke, far_pt[:, None, :]], axis=1)
    nrm = np.concatenate([normals, (-lookats)[:, None, :]], axis=1)
    return pts, nrm


# ----------------------------------------------------------------------------
# a4  point_inside_polyhedra
# ----------------------------------------------------------------------------
def point_inside_polyhedra(points, plane_points, normals):
    """intersections.py:146-164: inside <=> every plane has (p - pp).n >= 0."""
    P = np.asarray(points, np.float64)
    pp = np.asarray(plane_points, np.float64)
    nr = np.asarray(normals, np.float64)
    inside = np.ones(len(P), bool)
    for m in range(len(pp)):
        d0, d1, d2 = P[:, 0] - pp[m, 0], P[:, 1] - pp[m, 1], P[:, 2] - pp[m, 2]
        dp = (d0 * nr[m, 0] + d2 * nr[m, 2]) + d1 * nr[m, 1]         # see module docstring
        inside &= dp >= 0
    return inside


# ----------------------------------------------------------------------------
# a7  uv2pt scatter vote,  a8  segment
# ----------------------------------------------------------------------------
def vote_frame(votes, uv2pt, mask_flat):
    """One frame of VotingSegmentation.vote (voting.py:94-98), in place.

    ``votes[uv2pt[valid], mask[valid]] += 1`` is a buffered fancy-index add:
    a (point, label) pair that occurs several times in one frame adds 1, not
    its multiplicity (quirk Q1).  Negative lookups other than -1 wrap like
    NumPy indices; out-of-range point or label raises IndexError.
    """
    uv2pt = np.asarray(uv2pt)
    mask_flat = np.asarray(mask_flat)
    valid = uv2pt != -1
    if not valid.any():
        return votes
    pt = uv2pt[valid].astype(np.int64)
    lb = mask_flat[valid].astype(np.int64)
    n, c = votes.shape
    if ((pt >= n) | (pt < -n)).any() or (lb >= c).any():
        raise IndexError("vote index out of bounds")
    pt = np.where(pt < 0, pt + n, pt)
    flat = np.unique(pt * c + lb)
    votes.reshape(-1)[flat] += 1
    return votes


def segment(votes, nclasses, threshold=0.5, filter_classes=None):
    """VotingSegmentation.segment (voting.py:106-137) -> int64 [N].

    ``nclasses`` is the label written for "unclassified"; after a votes-file
    reload the reference passes votes.shape[1] here (quirk Q2).
    """
    votes = np.asarray(votes)
    n = len(votes)
    total = votes.sum(-1)                                             # :120 (all columns)
    sel = votes if filter_classes is None else votes[:, list(filter_classes)]
    cls = np.zeros(n, np.int64)
    best = np.zeros(n, votes.dtype)
    if sel.shape[1]:
        best = sel[:, 0].copy()
        for j in range(1, sel.shape[1]):                              # first maximum wins (:124)
            up = sel[:, j] > best
            cls[up] = j
            best[up] = sel[up, j]
    valid = total > 0
    cls[~valid] = nclasses                                            # :126
    with np.errstate(all='ignore'):
        low = valid & (best / np.where(valid, total, 1) < threshold)  # :128-130
    cls[low] = nclasses
    cls[best == 0] = nclasses                                         # :131
    if filter_classes is not None:
        for i, c in enumerate(filter_classes):                        # sequential, aliasing (Q3) :133-135
            cls[cls == i] = c
    return cls


def filter_remap_table(nclasses, filter_classes, size):
    """Value -> value table equivalent to the sequential loop at voting.py:133-135."""
    tab = np.arange(size, dtype=np.int64)
    if filter_classes is not None:
        for i, c in enumerate(filter_classes):
            tab[tab == i] = c
    return tab


# ----------------------------------------------------------------------------
# composed forward path (SURVEY 8(c) last row): project -> sample -> vote -> segment
# ----------------------------------------------------------------------------
def forward_votes(points, K, wxyzs, translations, masks, max_depth, w=None, h=None, out_dtype=np.float64, ncols=None):
    """Per view j: inside = point_inside_polyhedra(P, planes_j); uv = points2pixel(P[inside]);
    drop samples with u not in [0,W) or v not in [0,H); votes[idx, mask_j[v,u]] += 1.
    (fusion.py:254-266 + voting.py:94-98 with one sample per point per view.)"""
    P = np.asarray(points, np.float64)
    masks = np.asarray(masks)
    V, H, W = masks.shape
    w = W if w is None else w
    h = H if h is None else h
    ppts, pnrm = frustum_planes(K, w, h, wxyzs, translations, max_depth)
    if ncols is None:                                                   # VotingSegmentation allocates nclasses + 1 columns (voting.py:34)
        ncols = 134 if masks.dtype == np.uint8 else int(masks.max()) + 1
    votes = np.zeros((len(P), ncols), out_dtype)
    for j in range(V):
        inside = point_inside_polyhedra(P, ppts[j], pnrm[j])
        idx = np.nonzero(inside)[0]
        if not len(idx):
            continue
        u, v = points2pixel(P[idx], K, wxyzs[j], translations[j])
        ok = (u >= 0) & (u < W) & (v >= 0) & (v < H)
        idx, u, v = idx[ok], u[ok], v[ok]
        lab = masks[j, v, u].astype(np.int64)
        if (lab >= ncols).any():
            raise IndexError("mask label exceeds nclasses")
        votes[idx, lab] += 1                                            # idx is unique per view
    return votes


def project_vote_argmax(points, K, wxyzs, translations, masks, max_depth, nclasses=133,
                        threshold=0.5, filter_classes=None, return_votes=False):
    votes = forward_votes(points, K, wxyzs, translations, masks, max_depth, ncols=nclasses + 1)
    cls = segment(votes, nclasses, threshold, filter_classes)
    return (cls, votes) if return_votes else cls


# ----------------------------------------------------------------------------
# a9  mask post-processing of get2DSeg.SegmentImage
# ----------------------------------------------------------------------------
def sem_logits_to_mask(sem, conf_threshold=0.017, unclassified=133):
    """get2DSeg.py:110-118: argmax over classes; softmax-max < conf_threshold -> 133.

    float32 arithmetic like torch's softmax: exp(x - max) / sum; the maximum
    probability is 1 / sum(exp(x - max)).  Pinned by tests/golden/sem_mask.npz:
    the reference's own six statements run with CPU torch
    (tests/golden/make_golden_sem.py); labels are equal outside a 1e-5 relative
    band around the threshold (summation order of the 133 exponentials).  The
    network that produces the logits is third-party and absent.
    """
    sem = np.asarray(sem, np.float32)
    lab = sem.argmax(0).astype(np.int64)
    if conf_threshold:
        m = sem.max(0, keepdims=True)
        s = np.exp(sem - m, dtype=np.float32).sum(0, dtype=np.float32)
        pmax = np.float32(1) / s
        lab[pmax < np.float32(conf_threshold)] = unclassified
    return lab


# ----------------------------------------------------------------------------
# a10/a11  oriented boxes and merge_bb
# ----------------------------------------------------------------------------
def points_in_obb(points, center, R, extent):
    """open3d OrientedBoundingBox.get_point_indices_within_bounding_box as used at
    merge_intersecting_bb.py:76,87: |(p - c) . R[:, i]| <= extent_i / 2 for i = 0..2.
    open3d is absent: restated from its published algorithm, parity unpinned."""
    P = np.asarray(points, np.float64)
    c = np.asarray(center, np.float64)
    R = np.asarray(R, np.float64)
    e = np.asarray(extent, np.float64)
    d0, d1, d2 = P[:, 0] - c[0], P[:, 1] - c[1], P[:, 2] - c[2]
    ok = np.ones(len(P), bool)
    for i in range(3):
        pr = (d0 * R[0, i] + d1 * R[1, i]) + d2 * R[2, i]
        ok &= np.abs(pr) <= e[i] / 2
    return ok


def obb_from_points(pts):
    """open3d OrientedBoundingBox.create_from_points (merge_intersecting_bb.py:75,86,126;
    get3DSeg.py:434) restated from memory of its published algorithm: convex hull (Qhull) ->
    mean / covariance of the hull vertices -> eigenvectors sorted by descending eigenvalue,
    third axis = first x second -> extents from the min/max of the hull points in that frame.
    PARITY UNPINNED (open3d absent, no reference fixture)."""
    from scipy.spatial import ConvexHull
    pts = np.asarray(pts, np.float64)
    hull = ConvexHull(pts)
    hp = pts[np.sort(hull.vertices)]
    mean = hp.mean(0)
    cov = np.cov((hp - mean).T, bias=True)
    evals, evecs = np.linalg.eigh(cov)
    order = np.argsort(-evals, kind='stable')
    R = evecs[:, order]
    R[:, 2] = np.cross(R[:, 0], R[:, 1])
    loc = (hp - mean) @ R
    lo, hi = loc.min(0), loc.max(0)
    center = mean + R @ ((lo + hi) / 2)
    return center, R, hi - lo


def obb_corners(center, R, extent):
    """open3d get_box_points corner order (merge_intersecting_bb.py:127), restated; unpinned."""
    x, y, z = (R[:, i] * extent[i] / 2 for i in range(3))
    c = np.asarray(center)
    return np.array([c - x - y - z, c + x - y - z, c - x + y - z, c - x - y + z,
                     c + x + y + z, c - x + y + z, c + x - y + z, c + x + y - z])


def merge_bb(info_sem, ids, points, box_fn=obb_from_points):
    """merge_intersecting_bb.py:103-128 control flow (lists/arrays mutated in place like the
    reference), including its quirks: the list *index* is used as the point id (Q6), entries are
    deleted while the list shrinks (Q7), and check_intersection returns early when the partner
    has < 4 points (:83-84).  Returns (info_sem, ids)."""
    P = np.asarray(points, np.float64)
    n0 = len(info_sem)
    for id1 in range(1, n0):                                           # id_list is built once (:110-113)
        hits = []
        own = P[ids == id1]
        if len(own) >= 4:                                              # :72
            in1 = points_in_obb(P, *box_fn(own))
            for id2 in range(1, n0):
                if id1 != id2 and id2 < len(info_sem) - 1 and id1 < len(info_sem) - 1:   # :79
                    if info_sem[id1]["parent_id"] == info_sem[id2]["parent_id"]:
                        other = P[ids == id2]
                        if len(other) < 4:
                            break                                      # early return (:83-84)
                        in2 = points_in_obb(P, *box_fn(other))
                        if (in1 & in2).any():                          # :88-90
                            hits.append(id2)
        if hits:
            for b in hits:                                             # update_id_info (:58-62)
                sel = ids == b
                info_sem[id1]["area"] += info_sem[b]["area"]
                ids[sel] = id1
            for b in hits:
                if b < len(info_sem):
                    del info_sem[b]                                    # :118-120
    for k in range(1, len(info_sem)):                                  # :122-128
        own = P[ids == info_sem[k]["id"]]
        if len(own) > 4:
            info_sem[k]["bbox"] = obb_corners(*box_fn(own)).tolist()
    return info_sem, ids


# ----------------------------------------------------------------------------
# (f)#1  split_into_instances (segUtils/cv.py:402-500), literal restatement
# ----------------------------------------------------------------------------
def unproject_depth(depth, K, q_wxyz, t, depth_scale=1000):
    """RTAB2Cache.__getRGBP3d (RTAB_utils/ios_rtab.py:167-173) for one frame, then __getModP3d (:187-192): the NumPy
    expressions of those lines with the pinned ``rotate`` (a1) in place of SpatQuadranion.rotate.  No third-party
    arithmetic on this path; the file-reading around it (PIL, skimage.resize for the colours) is out of scope."""
    d = np.asarray(depth)
    H, W = d.shape
    px, py = np.meshgrid(np.linspace(0, W - 1, W), np.linspace(0, H - 1, H))              # :167-168
    cx = np.multiply(px - K[0, 2], d / K[0, 0])                                           # :171
    cy = np.multiply(py - K[1, 2], d / K[1, 1])                                           # :172
    pts = np.array([cx, cy, d]).transpose(1, 2, 0).reshape(-1, 3)                         # :173
    pts = np.divide(pts, depth_scale)                                                     # :187
    return rotate(q_wxyz, pts) + np.asarray(t, np.float64)                                # :190-192


def fuse_match_frame(uv, x_pts, x_nrm, x_clr, x_mrg, x_occ, ids, q_pts, q_nrm, q_clr, free, h, w, half, radius, min_cosine):
    """The per-frame matching loop of Fusion.fuse (fusion.py:269-298), literally: seeds in index order, each taking the free pixels
    of its window that pass the criterion (:223-228), arrays updated in place.  Returns the frame's uv2pt (int32 [h*w], -1 = none).
    Pinned through tests/golden/fuse.npz (the reference's own run); used to check the data-parallel formulation on random frames."""
    pcdimg = np.arange(h * w).reshape(h, w)
    pt2u, pt2v = np.arange(h * w) % w, np.arange(h * w) // w
    uv2pt = np.full(h * w, -1, np.int32)
    count = h * w
    for i_, (idx, (u_, v_)) in enumerate(zip(ids, np.asarray(uv).T)):
        if not count:
            break
        starti, endi = max(0, v_ - half), v_ + half + 1
        startj, endj = max(0, u_ - half), u_ + half + 1
        patch = pcdimg[starti:endi, startj:endj].reshape(-1)
        valid = free[starti:endi, startj:endj].reshape(-1)
        if not valid.any():
            continue
        patch = patch[valid]
        p_pts, p_nrm, p_clr = q_pts[patch], q_nrm[patch], q_clr[patch]
        dist = np.linalg.norm(p_pts - x_pts[i_][None, :], axis=-1)
        mask = (dist < radius) & (np.einsum('ij, j -> i', p_nrm, x_nrm[i_]) > min_cosine)
        matches = mask.sum()
        if matches:
            count -= matches
            x_pts[i_] = np.mean(np.vstack([p_pts[mask], x_pts[i_][None, :]]), axis=0)
            x_clr[i_] = np.mean(np.vstack([p_clr[mask], x_clr[i_][None, :]]), axis=0)
            n = np.mean(np.vstack([p_nrm[mask], x_nrm[i_][None, :]]), axis=0)
            x_nrm[i_] = n / np.linalg.norm(n)
            x_mrg[i_] += matches
            x_occ[i_] += 1
            merged = patch[mask]
            uv2pt[merged] = idx
            free[pt2v[merged], pt2u[merged]] = False
    return uv2pt


def radius_adjacency(points, r):
    """fusion.py:374-375: KDTree(points).query_radius(points, r) -- brute force over all pairs with the tree's leaf test:
    sklearn's euclidean_rdist accumulates (x1[j] - x2[j])**2 for j = 0, 1, 2 in that order and query_radius keeps
    rdist <= r**2 (sklearn/neighbors/_binary_tree.pxi, _dist_metrics.pyx; pinned against the installed sklearn by
    tests/test_oracle_golden.py).  Rows are returned sorted; sklearn's own order is the traversal's."""
    P = np.asarray(points, np.float64)
    r2 = float(r) * float(r)
    out = []
    for i in range(len(P)):
        t0, t1, t2 = P[i, 0] - P[:, 0], P[i, 1] - P[:, 1], P[i, 2] - P[:, 2]
        d = (t0 * t0 + t1 * t1) + t2 * t2
        out.append(np.nonzero(d <= r2)[0])
    return out


def split_into_instances(classes, adj, nclasses=133, instance_classes=None, minimum_points=1):
    """Flood fill from the lowest remaining index through same-class neighbours (directed neighbour lists, FIFO
    queue), clusters numbered in visit order, small ones folded into one bucket -- cv.py:425-500 line by line."""
    n = len(classes)
    classes = np.array(classes).copy()
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
        mask = classes == c
        remaining = np.nonzero(mask)[0]
        while len(remaining):
            seed = remaining[0]
            seed_class = classes[seed]
            inq = np.zeros(n, bool)
            inq[seed] = True
            queue, cluster, head = [seed], [], 0
            while head < len(queue):
                pnt = queue[head]; head += 1
                if classes[pnt] != seed_class:
                    continue
                cluster.append(pnt)
                new = [q for q in adj[pnt] if not inq[q]]
                queue += new
                inq[new] = True
            cluster = np.array(cluster)
            if len(cluster) < minimum_points:
                cat = nclasses
                if small_id is None:
                    small_id = ninst
                    info.append({'id': ninst, 'isthing': True, 'category_id': int(cat), 'area': 0})
                    ninst += 1
                info[small_id]['area'] += len(cluster)
                ids[cluster] = small_id
            else:
                cat = c
                info.append({'id': ninst, 'isthing': True, 'category_id': int(cat), 'area': int(len(cluster))})
                ids[cluster] = ninst
                ninst += 1
            mask[cluster] = False
            remaining = np.nonzero(mask)[0]
            classes[cluster] = cat
    return np.arange(ninst), ids, info, classes
