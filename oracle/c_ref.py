"""ctypes loader of the C oracle (oracle/f3d_oracle.c).  TEST INFRASTRUCTURE, NOT PRODUCT (see np_ref.py)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / '_build' / 'libf3d_oracle.so'
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.is_file() or _SO.stat().st_mtime < (_HERE / 'f3d_oracle.c').stat().st_mtime:
            subprocess.run(['make', '-C', str(_HERE)], check=True, capture_output=True)
        _lib = C.CDLL(str(_SO))
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rotate(q, pts):
    pts = _d(pts); out = np.empty_like(pts)
    lib().orc_rotate(_p(_d(q)), _p(pts), C.c_int64(len(pts)), _p(out))
    return out


def points2pixel(pts, K, q, t):
    pts = _d(pts); uv = np.empty((2, len(pts)), np.int32)
    rc = lib().orc_points2pixel(_p(pts), C.c_int64(len(pts)), _p(_d(K)), _p(_d(q)), _p(_d(t)), _p(uv))
    if rc:
        raise ZeroDivisionError('zero quaternion')
    return uv


def frustum_planes(K, w, h, q, t, max_depth):
    q, t = _d(np.atleast_2d(q)), _d(np.atleast_2d(t))
    V = len(t)
    pp, pn, eyes, look = np.empty((V, 5, 3)), np.empty((V, 5, 3)), np.empty((V, 3)), np.empty((V, 3))
    lib().orc_frustum_planes(_p(_d(K)), C.c_double(w), C.c_double(h), _p(q), _p(t), C.c_int(V), C.c_double(max_depth),
                             _p(pp), _p(pn), _p(eyes), _p(look))
    return pp, pn, eyes, look


def point_inside_polyhedra(pts, pp, nr):
    pts, pp, nr = _d(pts), _d(pp), _d(nr)
    out = np.empty(len(pts), np.uint8)
    lib().orc_inside_polyhedra(_p(pts), C.c_int64(len(pts)), _p(pp), _p(nr), C.c_int(len(pp)), _p(out))
    return out.view(np.bool_)


def vote_frame(votes, uv2pt, mask_flat):
    lut = np.ascontiguousarray(uv2pt, np.int32); m = np.ascontiguousarray(mask_flat, np.uint8)
    rc = lib().orc_vote_frame(_p(votes), C.c_int64(votes.shape[0]), C.c_int(votes.shape[1]), _p(lut), _p(m), C.c_int64(len(lut)))
    if rc:
        raise IndexError('vote index out of bounds')
    return votes


def segment(votes, nclasses, threshold=0.5, filter_classes=None):
    votes = _d(votes)
    f = None if filter_classes is None else np.ascontiguousarray(list(filter_classes), np.int32)
    out = np.empty(len(votes), np.int64)
    lib().orc_segment(_p(votes), C.c_int64(len(votes)), C.c_int(votes.shape[1]), C.c_int(nclasses), C.c_double(threshold),
                      _p(f), C.c_int(0 if f is None else len(f)), _p(out))
    return out


def project_vote_argmax(pts, K, q, t, masks, max_depth, nclasses=133, threshold=0.5, filter_classes=None, return_votes=False):
    pts, q, t = _d(pts), _d(q), _d(t)
    masks = np.ascontiguousarray(masks, np.uint8)
    V, H, W = masks.shape
    f = None if filter_classes is None else np.ascontiguousarray(list(filter_classes), np.int32)
    cls = np.empty(len(pts), np.int64)
    votes = np.zeros((len(pts), nclasses + 1)) if return_votes else None
    rc = lib().orc_project_vote_argmax(_p(pts), C.c_int64(len(pts)), _p(_d(K)), C.c_double(W), C.c_double(H), _p(q), _p(t), C.c_int(V),
                                       C.c_double(max_depth), _p(masks), C.c_int(H), C.c_int(W), C.c_int(nclasses),
                                       C.c_double(threshold), _p(f), C.c_int(0 if f is None else len(f)), _p(cls), _p(votes))
    if rc == -3:
        raise IndexError('mask label exceeds nclasses')
    if rc:
        raise ZeroDivisionError('zero quaternion')
    return (cls, votes) if return_votes else cls
