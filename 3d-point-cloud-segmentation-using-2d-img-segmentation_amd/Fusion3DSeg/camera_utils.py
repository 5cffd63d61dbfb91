"""Drop-in for the reference's Fusion3DSeg/camera_utils.py.

``points2pixel`` is the hot function: it runs as a HIP kernel (f3d_points2pixel_f64).  The frustum
helpers act on a handful of points per camera and stay on the host; the per-view plane records the
kernels consume come from ``f3d.views_build`` (C++ in libf3d_hip.so), which follows the same steps.
"""
import numpy as np

import f3d
from RTAB_utils.spatQuad import SpatQuadranion


def points2pixel(points, intrinsic, quat, translation):
    """World points [N,3] -> int32 [2,N] pixel (u=column, v=row) of the camera (quat wxyz, translation).
    Same contract as reference camera_utils.py:9-26: no z>0 test, no image-bounds test."""
    return f3d.default_context().points2pixel(points, intrinsic, quat, translation)


def pixel2point(u, v, K, R, eye):
    ray = np.linalg.inv(K) @ np.array([u, v, 1])
    return R @ ray + eye


def get_camera_frustum(K, width, height):
    """Eye, the four image corners at depth 1 and the principal ray, in camera coordinates ([6,3]), plus
    the 9 frustum edges (reference camera_utils.py:60-93)."""
    pix = np.array([[0, 0, 0], [0, 0, 1], [width, 0, 1], [width, height, 1], [0, height, 1],
                    [width / 2, height / 2, 1]], np.float64)
    edges = np.array([[0, 1], [0, 2], [0, 3], [0, 4], [1, 2], [2, 3], [3, 4], [4, 1], [0, 5]])
    return (np.linalg.inv(K) @ pix.T).T, edges


def camera2world(frame_points, xyzws, translations, rescale=1000):
    """Camera-frame points of F frames -> world frame ([F,N,3]); broadcasting rules of reference :96-132."""
    pts = np.asarray(frame_points, np.float64) / rescale
    qs, ts = np.asarray(xyzws, np.float64), np.asarray(translations, np.float64)
    F = max(len(pts) if pts.ndim > 2 else 1, len(qs) if qs.ndim > 1 else 1, len(ts) if ts.ndim > 1 else 1)
    pts = np.broadcast_to(pts, (F,) + pts.shape[-2:])
    qs = np.broadcast_to(qs, (F, 4))
    ts = np.broadcast_to(ts, (F, 3))
    return np.array([SpatQuadranion(q).rotate(p) + t for p, q, t in zip(pts, qs, ts)])


def get_frustum_unit_vectors(frustum_points):
    eyes = frustum_points[:, 0, :]
    rays = frustum_points[:, 1:, :] - eyes[:, None, :]
    rays = rays / np.linalg.norm(rays, axis=-1)[..., None]
    return eyes, rays[:, :-1, :], rays[:, -1, :]


def get_frustum_face_normals(eyes, corners):
    a = corners - eyes[:, None, :]
    b = np.roll(corners, -1, axis=1) - eyes[:, None, :]
    n = np.cross(a, b)
    return n / np.linalg.norm(n, axis=-1)[..., None]
