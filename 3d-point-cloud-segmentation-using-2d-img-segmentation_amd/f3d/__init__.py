"""ctypes binding of libf3d_hip.so (include/f3d.h) -- the only way Python reaches the kernels.

There is deliberately no CPU fallback here: if the library is missing or no HIP
device is usable, every compute call raises ``F3DUnavailable``.

Two call styles, mirroring the C-ABI:
* NumPy in / NumPy out (host-pointer entry points) -- what the drop-in modules
  ``Fusion3DSeg.*`` use;
* ``*_dev`` methods taking raw device pointers (``tensor.data_ptr()``) and a
  stream handle -- what ``bench.py`` and device-resident pipelines use.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

__all__ = ['F3DError', 'F3DUnavailable', 'Context', 'default_context', 'library', 'library_path',
           'views_build', 'frustum_data', 'quat_inverse', 'VIEW_DOUBLES', 'F64', 'F32', 'FUSE_SORT', 'FUSE_GATHER']

F64, F32 = 0, 1
FUSE_SORT, FUSE_GATHER = 2, 4
OK, ERR_INVALID, ERR_HIP, ERR_INDEX, ERR_ZERO_QUAT, ERR_NOMEM = 0, -1, -2, -3, -4, -5
VIEW_DOUBLES = 88            # sizeof(f3d_view) / 8
OBB_DOUBLES = 15             # sizeof(f3d_obb) / 8
MAX_OBB = 4096
OBB_OK, OBB_FEW, OBB_DEFERRED = 0, 1, 2


class F3DError(RuntimeError):
    pass


class F3DUnavailable(F3DError):
    """libf3d_hip.so is not built/loadable or there is no HIP device."""


_lib = None


def library_path():
    """The in-tree build.  F3D_LIBRARY names another build of the same library (the A/B timing scripts under scripts/ point the
    loader at a variant this way instead of overwriting the product's file)."""
    alt = os.environ.get('F3D_LIBRARY')
    return Path(alt).resolve() if alt else Path(__file__).resolve().parent / 'libf3d_hip.so'


def library():
    """Load libf3d_hip.so once and declare every prototype of include/f3d.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.is_file():
        raise F3DUnavailable(f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                             f'(or `make -C {path.parent.parent / "csrc"}`); there is no CPU fallback')
    try:                         # share torch's HIP runtime when torch is in the process (same SONAME)
        import torch  # noqa: F401
    except Exception:            # torch is plumbing, not a requirement of the binding
        pass
    try:
        lib = C.CDLL(str(path), mode=getattr(os, 'RTLD_NOW', 2))
    except OSError as exc:
        raise F3DUnavailable(f'cannot load {path}: {exc}') from exc

    vp, i32, i64, dbl, flt = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
    protos = {
        'f3d_version': (i32, []),
        'f3d_ctx_create': (vp, [i32]),
        'f3d_ctx_destroy': (None, [vp]),
        'f3d_last_error': (C.c_char_p, [vp]),
        'f3d_ctx_synchronize': (i32, [vp]),
        'f3d_ctx_stream': (vp, [vp]),
        'f3d_ctx_reserve': (i32, [vp, i64, i32, i32, i32]),
        'f3d_ctx_set_strict': (i32, [vp, i32]),
        'f3d_ctx_alloc_count': (C.c_longlong, [vp]),
        'f3d_quat_inverse': (i32, [vp, vp]),
        'f3d_frustum_data': (i32, [vp, dbl, dbl, vp, vp, i32, vp, vp, vp]),
        'f3d_views_build': (i32, [vp, dbl, dbl, vp, vp, i32, dbl, vp]),
        'f3d_rotate_f64': (i32, [vp, vp, i64, vp, vp]),
        'f3d_rotate_f64_dev': (i32, [vp, vp, i64, vp, vp, vp]),
        'f3d_points2pixel_f64': (i32, [vp, vp, i64, vp, vp, vp, vp]),
        'f3d_points2pixel_dev': (i32, [vp, vp, i32, i64, vp, vp, vp, vp, vp]),
        'f3d_inside_polyhedra_f64': (i32, [vp, vp, i64, vp, vp, i32, vp]),
        'f3d_inside_polyhedra_dev': (i32, [vp, vp, i32, i64, vp, vp, i32, vp, vp]),
        'f3d_project_view_f64': (i32, [vp, vp, i64, vp, vp, vp]),
        'f3d_project_view_dev': (i32, [vp, vp, i32, i64, vp, vp, vp, vp]),
        'f3d_project_vote_argmax': (i32, [vp, vp, i32, i64, vp, i32, vp, i32, i32, i32, vp, i32, dbl, vp, vp]),
        'f3d_project_vote_argmax_dev': (i32, [vp, vp, i32, i64, vp, i32, vp, i32, i32, i32, vp, i32, dbl, vp, vp, C.c_uint, vp, vp]),
        'f3d_mask_presence_dev': (i32, [vp, vp, i32, i32, i32, vp, vp]),
        'f3d_fuse_chunked_begin_dev': (i32, [vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]),
        'f3d_fuse_chunk_dev': (i32, [vp, vp, i32, i64, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, dbl, vp, C.c_uint, vp, vp]),
        'f3d_coded_plane_bytes': (C.c_size_t, [i32, i32]),
        'f3d_code_planes_dev': (i32, [vp, vp, i32, i32, i32, vp, vp]),
        'f3d_fuse_chunk_coded_dev': (i32, [vp, vp, i32, i64, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, dbl, vp, C.c_uint, vp, vp]),
        'f3d_debug_fastpath_audit': (i32, [vp, vp, i32, i64, vp, i32, i32, i32, vp]),
        'f3d_debug_fuse_deferred': (i32, [vp, vp, vp]),
        'f3d_cloud_sort_cells_dev': (i32, [vp, vp, i32, i64, vp, vp, vp]),
        'f3d_take_device_error': (i32, [vp, vp]),
        'f3d_vote_uv2pt': (i32, [vp, vp, vp, i64, vp, i64, i32]),
        'f3d_vote_uv2pt_dev': (i32, [vp, vp, vp, i64, vp, i64, i32, vp]),
        'f3d_vote_uv2pt_batch': (i32, [vp, vp, vp, i64, i32, i32, vp, i64, i32]),
        'f3d_vote_uv2pt_batch_dev': (i32, [vp, vp, vp, i64, i32, i32, vp, i64, i32, vp]),
        'f3d_segment_votes': (i32, [vp, vp, i64, i32, i32, dbl, vp, i32, vp]),
        'f3d_segment_votes_dev': (i32, [vp, vp, i64, i32, i32, dbl, vp, i32, vp, vp]),
        'f3d_sem_logits_to_mask': (i32, [vp, vp, i32, i64, flt, i32, vp]),
        'f3d_sem_logits_to_mask_dev': (i32, [vp, vp, i32, i64, flt, i32, vp, vp]),
        'f3d_sem_logits_to_masks_dev': (i32, [vp, vp, i32, i32, i64, flt, i32, vp, vp]),
        'f3d_points_in_obb': (i32, [vp, vp, i32, i64, vp, i32, vp, vp]),
        'f3d_points_in_obb_dev': (i32, [vp, vp, i32, i64, vp, i32, vp, vp, vp]),
        'f3d_group_by_id': (i32, [vp, vp, i64, i64, vp, vp]),
        'f3d_obb_extremes': (i32, [vp, vp, i32, i64, vp]),
        'f3d_obb_hull_filter': (i32, [vp, i64, vp, vp, vp, vp, vp]),
        'f3d_group_by_id_dev': (i32, [vp, vp, i64, i64, vp, vp, vp, vp]),
        'f3d_obb_extremes_dev': (i32, [vp, vp, i32, i64, vp, vp, i64, vp, vp]),
        'f3d_obb_hull_filter_dev': (i32, [vp, vp, i32, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]),
        'f3d_obb_fit': (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
        'f3d_obb_fit_dev': (i32, [vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]),
        'f3d_obb_candidates_dev': (i32, [vp, vp, i32, i64, vp, vp, vp, i64, i32, vp, vp, vp]),
        'f3d_gather_points_dev': (i32, [vp, vp, i32, vp, i64, vp, vp]),
        'f3d_relabel': (i32, [vp, vp, i64, i64, i64, vp]),
        'f3d_relabel_dev': (i32, [vp, vp, i64, i64, i64, vp, vp]),
        'f3d_ray_x_lines': (i32, [vp, vp, vp, vp, vp, i64, vp, vp]),
        'f3d_rays_x_plane': (i32, [vp, vp, vp, vp, vp, i64, vp, vp]),
        'f3d_lines_x_planes': (i32, [vp, vp, vp, i64, vp, vp, i32, vp, vp]),
        'f3d_point_inside_polygon': (i32, [vp, vp, i64, vp, i32, vp, vp]),
        'f3d_points_plane_projection': (i32, [vp, vp, i64, vp, vp, vp]),
        'f3d_lines_plane_projection': (i32, [vp, vp, vp, i64, vp, vp, vp, vp, vp]),
        'f3d_components_same_class': (i32, [vp, vp, i64, vp, vp, vp]),
        'f3d_components_same_class_dev': (i32, [vp, vp, i64, vp, vp, vp, vp, vp]),
        'f3d_patch_owner': (i32, [vp, vp, i64, i32, i32, i32, dbl, dbl, vp, vp, vp, vp, vp, vp]),
        'f3d_patch_owner_dev': (i32, [vp, vp, i64, i32, i32, i32, dbl, dbl, vp, vp, vp, vp, vp, vp, vp]),
        'f3d_patch_seeds': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, dbl, dbl, vp, vp]),
        'f3d_patch_match': (i32, [vp, vp, i64, i32, i32, i32, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        'f3d_patch_seeds_sums': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, dbl, dbl, vp, vp, vp, vp]),
        'f3d_unproject_depth': (i32, [vp, vp, i32, i32, i32, vp, dbl, vp, vp, vp]),
        'f3d_unproject_depth_dev': (i32, [vp, vp, i32, i32, i32, vp, dbl, vp, vp, vp, vp]),
        'f3d_unproject_depth_batch_dev': (i32, [vp, vp, i32, i32, i32, i32, vp, dbl, vp, vp, vp, vp]),
        'f3d_radius_graph_count': (i32, [vp, vp, i32, i64, dbl, vp, vp]),
        'f3d_radius_graph_fill': (i32, [vp, i64, vp]),
        'f3d_radius_graph_count_dev': (i32, [vp, vp, i32, i64, dbl, vp, vp, vp]),
        'f3d_radius_graph_fill_dev': (i32, [vp, i64, vp, vp, vp]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    lib._f3d_symbols = tuple(protos)
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f'expected shape {tuple(shape)}, got {a.shape}')
    return a


def _raise(code, msg):
    if code == ERR_INDEX:
        raise IndexError(msg)
    if code == ERR_ZERO_QUAT:
        raise ZeroDivisionError(msg)
    if code == ERR_INVALID:
        raise ValueError(msg)
    if code == ERR_NOMEM:
        raise MemoryError(msg)
    if code == ERR_HIP:
        raise F3DUnavailable(msg)
    raise F3DError(f'f3d error {code}: {msg}')


# ------------------------------------------------------------------ host geometry (no device)
def quat_inverse(q_wxyz):
    q = _f64(q_wxyz, (4,))
    out = np.empty(4)
    rc = library().f3d_quat_inverse(_ptr(q), _ptr(out))
    if rc:
        _raise(rc, 'a zero quaternion cannot be inverted')
    return out


def frustum_data(K, w, h, wxyzs, translations):
    """eyes [V,3], lookats [V,3], face_normals [V,4,3] of Fusion._get_frustum_data (fusion.py:119-132)."""
    K = _f64(K, (3, 3))
    q = _f64(np.atleast_2d(wxyzs))
    t = _f64(np.atleast_2d(translations))
    V = len(t)
    if q.shape != (V, 4) or t.shape != (V, 3):
        raise ValueError('wxyzs must be [V,4] and translations [V,3]')
    eyes, look, nrm = np.empty((V, 3)), np.empty((V, 3)), np.empty((V, 4, 3))
    rc = library().f3d_frustum_data(_ptr(K), float(w), float(h), _ptr(q), _ptr(t), V, _ptr(eyes), _ptr(look), _ptr(nrm))
    if rc:
        _raise(rc, 'f3d_frustum_data failed')
    return eyes, look, nrm


def views_build(K, w, h, wxyzs, translations, max_depth):
    """Packed per-view records (704 bytes each, viewed as float64 [V, 88]) consumed by the fused kernels."""
    K = _f64(K, (3, 3))
    q = _f64(np.atleast_2d(wxyzs))
    t = _f64(np.atleast_2d(translations))
    V = len(t)
    if q.shape != (V, 4) or t.shape != (V, 3):
        raise ValueError('wxyzs must be [V,4] and translations [V,3]')
    out = np.zeros((V, VIEW_DOUBLES))
    rc = library().f3d_views_build(_ptr(K), float(w), float(h), _ptr(q), _ptr(t), V, float(max_depth), _ptr(out))
    if rc:
        _raise(rc, 'a zero quaternion cannot be inverted' if rc == ERR_ZERO_QUAT else 'f3d_views_build failed')
    return out


def view_fields(views):
    """Named sub-arrays of a [V,88] view table (for tests and debugging)."""
    v = np.asarray(views)
    return {'M': v[:, 0:9].reshape(-1, 3, 3), 't': v[:, 9:12], 'mnorm': v[:, 12:15],
            'K': v[:, 40:49].reshape(-1, 3, 3), 'qinv': v[:, 49:53],
            'plane_pt': v[:, 53:68].reshape(-1, 5, 3), 'plane_n': v[:, 68:83].reshape(-1, 5, 3), 'plane_off': v[:, 83:88]}


def _xyz(points):
    p = np.asarray(points)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError(f'points must be [N,3], got {p.shape}')
    if p.dtype == np.float32:
        return np.ascontiguousarray(p), F32
    return np.ascontiguousarray(p, dtype=np.float64), F64


def _filter(filter_classes):
    if filter_classes is None:
        return None, 0
    f = np.ascontiguousarray(np.asarray(list(filter_classes)), dtype=np.int32)
    return f, len(f)


class Context:
    """One f3d_ctx (device ordinal, stream, scratch arena).  Not thread-safe."""

    def __init__(self, device=0):
        self._lib = library()
        self._h = self._lib.f3d_ctx_create(int(device))
        if not self._h:
            raise F3DUnavailable(self._lib.f3d_last_error(None).decode() or 'f3d_ctx_create failed')
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.f3d_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            _raise(rc, self._lib.f3d_last_error(self._h).decode())

    @property
    def stream(self):
        return self._lib.f3d_ctx_stream(self._h)

    def synchronize(self):
        self._check(self._lib.f3d_ctx_synchronize(self._h))

    def reserve(self, n=0, nviews=0, h=0, w=0):
        """Size the scratch of the fused path / cell sort / uv2pt vote beforehand: later _dev calls do not allocate."""
        self._check(self._lib.f3d_ctx_reserve(self._h, int(n), int(nviews), int(h), int(w)))

    def set_strict(self, strict=True):
        self._check(self._lib.f3d_ctx_set_strict(self._h, int(bool(strict))))

    @property
    def alloc_count(self):
        return int(self._lib.f3d_ctx_alloc_count(self._h))

    # ---------------------------------------------------------------- NumPy (host pointer) calls
    def rotate(self, points, q_wxyz):
        p = _f64(points)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError('points must be [N,3]')
        q = _f64(q_wxyz, (4,))
        out = np.empty_like(p)
        self._check(self._lib.f3d_rotate_f64(self._h, _ptr(p), len(p), _ptr(q), _ptr(out)))
        return out

    def points2pixel(self, points, intrinsic, quat, translation):
        p = _f64(points)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError('points must be [N,3]')
        K, q, t = _f64(intrinsic, (3, 3)), _f64(quat, (4,)), _f64(translation, (3,))
        uv = np.empty((2, len(p)), np.int32)
        self._check(self._lib.f3d_points2pixel_f64(self._h, _ptr(p), len(p), _ptr(K), _ptr(q), _ptr(t), _ptr(uv)))
        return uv

    def inside_polyhedra(self, points, plane_points, normals):
        p = _f64(points)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError('points must be [N,3]')
        pp, nr = _f64(plane_points), _f64(normals)
        if pp.shape != nr.shape or pp.ndim != 2 or pp.shape[1] != 3:
            raise ValueError('plane_points and normals must both be [M,3]')
        out = np.empty(len(p), np.uint8)
        self._check(self._lib.f3d_inside_polyhedra_f64(self._h, _ptr(p), len(p), _ptr(pp), _ptr(nr), len(pp), _ptr(out)))
        return out.view(np.bool_)

    def project_view(self, points, view, want_uv=True, want_inside=True):
        p = _f64(points)
        v = _f64(view, (VIEW_DOUBLES,))
        uv = np.empty((2, len(p)), np.int32) if want_uv else None
        ins = np.empty(len(p), np.uint8) if want_inside else None
        self._check(self._lib.f3d_project_view_f64(self._h, _ptr(p), len(p), _ptr(v), _ptr(uv), _ptr(ins)))
        return uv, (None if ins is None else ins.view(np.bool_))

    def project_vote_argmax(self, points, views, masks, nclasses=133, threshold=0.5, filter_classes=None,
                            return_votes=False):
        p, dt = _xyz(points)
        views = _f64(views)
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        if masks.ndim != 3 or views.ndim != 2 or views.shape[1] != VIEW_DOUBLES or len(views) != len(masks):
            raise ValueError(f'views must be [V,{VIEW_DOUBLES}] and masks uint8 [V,H,W]')
        V, H, W = masks.shape
        f, nf = _filter(filter_classes)
        cls = np.empty(len(p), np.int64)
        votes = np.empty((len(p), nclasses + 1), np.uint16) if return_votes else None
        self._check(self._lib.f3d_project_vote_argmax(self._h, _ptr(p), dt, len(p), _ptr(views), V, _ptr(masks), H, W,
                                                      int(nclasses), _ptr(f), nf, float(threshold), _ptr(cls), _ptr(votes)))
        return (cls, votes) if return_votes else cls

    def vote_uv2pt(self, votes, uv2pt, mask_flat):
        """In-place on `votes` (float64 [npts, ncols], C-contiguous), like voting.py:98."""
        if votes.dtype != np.float64 or not votes.flags.c_contiguous or votes.ndim != 2:
            raise ValueError('votes must be a C-contiguous float64 [npts, ncols] array')
        lut = np.ascontiguousarray(uv2pt, dtype=np.int32).reshape(-1)
        m = np.ascontiguousarray(mask_flat, dtype=np.uint8).reshape(-1)
        if len(lut) != len(m):
            raise IndexError(f'shape mismatch: uv2pt has {len(lut)} entries, mask {len(m)}')
        self._check(self._lib.f3d_vote_uv2pt(self._h, _ptr(lut), _ptr(m), len(lut), _ptr(votes), votes.shape[0], votes.shape[1]))
        return votes

    def vote_uv2pt_batch(self, votes, luts, masks, h, w):
        """All frames of VotingSegmentation.vote in one call, in place on `votes` (float64 [npts, ncols], C-contiguous);
        luts int32 [F, h*w], masks uint8 [F, h*w]."""
        if votes.dtype != np.float64 or not votes.flags.c_contiguous or votes.ndim != 2:
            raise ValueError('votes must be a C-contiguous float64 [npts, ncols] array')
        lut = np.ascontiguousarray(luts, dtype=np.int32).reshape(-1, h * w)
        m = np.ascontiguousarray(masks, dtype=np.uint8).reshape(-1, h * w)
        if lut.shape != m.shape:
            raise IndexError(f'shape mismatch: lookups {lut.shape}, masks {m.shape}')
        self._check(self._lib.f3d_vote_uv2pt_batch(self._h, _ptr(lut), _ptr(m), len(lut), int(h), int(w), _ptr(votes), votes.shape[0], votes.shape[1]))
        return votes

    def segment_votes(self, votes, nclasses, threshold=0.5, filter_classes=None):
        v = _f64(votes)
        if v.ndim != 2:
            raise ValueError('votes must be [npts, ncols]')
        f, nf = _filter(filter_classes)
        cls = np.empty(len(v), np.int64)
        if v.shape[1] == 0:
            raise ValueError('attempt to get argmax of an empty sequence')
        self._check(self._lib.f3d_segment_votes(self._h, _ptr(v), v.shape[0], v.shape[1], int(nclasses), float(threshold),
                                                _ptr(f), nf, _ptr(cls)))
        return cls

    def sem_logits_to_mask(self, sem, conf_threshold=0.017, low_label=133):
        s = np.ascontiguousarray(sem, dtype=np.float32)
        if s.ndim != 3:
            raise ValueError('sem must be [C,H,W]')
        c, h, w = s.shape
        out = np.empty((h, w), np.uint8)
        self._check(self._lib.f3d_sem_logits_to_mask(self._h, _ptr(s), c, h * w, float(conf_threshold or 0.0), int(low_label), _ptr(out)))
        return out

    def points_in_obb(self, points, boxes, want_bits=True, want_cooc=True):
        """boxes: float64 [B,15] = center(3), R row-major(9), extent(3).  Returns (inside bool [N,B] or None, cooc bool [B,B] or None)."""
        p, dt = _xyz(points)
        b = _f64(boxes)
        if b.ndim != 2 or b.shape[1] != OBB_DOUBLES:
            raise ValueError('boxes must be [B,15]')
        B = len(b)
        words = (B + 31) // 32
        bits = np.zeros((len(p), words), np.uint32) if want_bits else None
        cooc = np.zeros((B, B), np.uint8) if want_cooc else None
        if B:
            self._check(self._lib.f3d_points_in_obb(self._h, _ptr(p), dt, len(p), _ptr(b), B, _ptr(bits), _ptr(cooc)))
        inside = None
        if want_bits:
            inside = np.unpackbits(bits.view(np.uint8), axis=1, bitorder='little')[:, :B].astype(bool)
        return inside, (None if cooc is None else cooc.astype(bool))

    # ---- per-instance point lists and hull candidates (merge_bb's box fits); host-pointer sequence, state kept in the context
    def group_by_id(self, ids, nids):
        """order int32 [n] (members of id 0, 1, ... in ascending point index; ids outside [0, nids) last), starts int64 [nids + 2]."""
        i = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        order = np.empty(len(i), np.int32)
        starts = np.empty(int(nids) + 2, np.int64)
        self._check(self._lib.f3d_group_by_id(self._h, _ptr(i), len(i), int(nids), _ptr(order), _ptr(starts)))
        self._grp = (len(i), int(nids))
        return order, starts

    def obb_extremes(self, points):
        """int32 [nids, 26]: per id the member extreme along +-x, +-y, +-z and the face / body diagonals (-1: no members).
        Follows group_by_id of the same cloud."""
        p, dt = _xyz(points)
        n, nids = self._grp
        out = np.empty((nids, 26), np.int32)
        self._check(self._lib.f3d_obb_extremes(self._h, _ptr(p), dt, len(p), _ptr(out)))
        return out

    def obb_hull_filter(self, facet_start, facets, margin):
        """Members not strictly inside their id's polytope: (cand int32 [n] grouped like `order`, cand_count int32 [nids]).
        Follows group_by_id and obb_extremes of the same cloud."""
        n, nids = self._grp
        fs = np.ascontiguousarray(facet_start, dtype=np.int32)
        eq = np.ascontiguousarray(facets, dtype=np.float64).reshape(-1, 4)
        mg = np.ascontiguousarray(margin, dtype=np.float64)
        if len(fs) != nids + 1 or len(mg) != nids or fs[-1] != len(eq):
            raise ValueError('facet_start must have nids + 1 entries ending at len(facets); margin one entry per id')
        cand = np.empty(n, np.int32)
        cnt = np.empty(nids, np.int32)
        self._check(self._lib.f3d_obb_hull_filter(self._h, n, _ptr(fs), _ptr(eq), _ptr(mg), _ptr(cand), _ptr(cnt)))
        return cand, cnt

    def obb_fit(self, point_sets, want_vertices=False):
        """Oriented boxes (Open3D's create_from_points recipe) of a list of point sets in ONE launch: boxes float64 [k, 15] (center,
        R row-major with the axes as columns, extent), status int32 [k] (OBB_OK / OBB_FEW / OBB_DEFERRED: fit that one on the host);
        with want_vertices also a list of bool arrays marking every set's hull vertices."""
        sets = [_f64(np.asarray(p).reshape(-1, 3)) for p in point_sets]
        start = np.zeros(len(sets) + 1, np.int64)
        start[1:] = np.cumsum([len(p) for p in sets])
        pts = np.ascontiguousarray(np.concatenate(sets)) if sets and start[-1] else np.zeros((0, 3))
        boxes = np.zeros((len(sets), OBB_DOUBLES))
        status = np.zeros(len(sets), np.int32)
        isvert = np.zeros(len(pts), np.uint8) if want_vertices else None
        self._check(self._lib.f3d_obb_fit(self._h, _ptr(pts), _ptr(start), len(sets), _ptr(boxes), _ptr(status), _ptr(isvert), None))
        if want_vertices:
            return boxes, status, [isvert[start[k]:start[k + 1]].view(np.bool_) for k in range(len(sets))]
        return boxes, status

    def relabel(self, ids, from_id, to_id):
        if ids.dtype != np.int64 or not ids.flags.c_contiguous:
            raise ValueError('ids must be a C-contiguous int64 array')
        cnt = np.zeros(1, np.int64)
        self._check(self._lib.f3d_relabel(self._h, _ptr(ids), ids.size, int(from_id), int(to_id), _ptr(cnt)))
        return int(cnt[0])

    # ---- the other intersections.py primitives (a12)
    @staticmethod
    def _n3(a):
        a = _f64(a)
        if a.ndim != 2 or a.shape[1] != 3:
            raise ValueError(f'expected [N,3], got {a.shape}')
        return a

    def ray_x_lines(self, origin, direction, starts, ends):
        o, d, s, e = _f64(origin, (3,)), _f64(direction, (3,)), self._n3(starts), self._n3(ends)
        pts, within = np.empty_like(s), np.empty(len(s), np.uint8)
        self._check(self._lib.f3d_ray_x_lines(self._h, _ptr(o), _ptr(d), _ptr(s), _ptr(e), len(s), _ptr(pts), _ptr(within)))
        return pts, within.view(np.bool_)

    def rays_x_plane(self, plane_point, plane_normal, origins, directions):
        pp, pn, o, d = _f64(plane_point, (3,)), _f64(plane_normal, (3,)), self._n3(origins), self._n3(directions)
        pts, valid = np.empty_like(o), np.empty(len(o), np.uint8)
        self._check(self._lib.f3d_rays_x_plane(self._h, _ptr(pp), _ptr(pn), _ptr(o), _ptr(d), len(o), _ptr(pts), _ptr(valid)))
        return pts, valid.view(np.bool_)

    def lines_x_planes(self, line_origins, line_ends, plane_points, plane_normals):
        lo, le, pp, pn = self._n3(line_origins), self._n3(line_ends), self._n3(plane_points), self._n3(plane_normals)
        pts, valid = np.empty((len(lo), len(pp), 3)), np.empty((len(lo), len(pp)), np.uint8)
        self._check(self._lib.f3d_lines_x_planes(self._h, _ptr(lo), _ptr(le), len(lo), _ptr(pp), _ptr(pn), len(pp), _ptr(pts), _ptr(valid)))
        return pts, valid.view(np.bool_)

    def point_inside_polygon(self, points, vertices):
        p, v = self._n3(points), self._n3(vertices)
        inside, within = np.empty(len(p), np.uint8), np.empty((len(v), len(p)), np.uint8)
        self._check(self._lib.f3d_point_inside_polygon(self._h, _ptr(p), len(p), _ptr(v), len(v), _ptr(inside), _ptr(within)))
        return inside.view(np.bool_), within.view(np.bool_)

    def points_plane_projection(self, points, plane_point, normal):
        p, pp, nr = self._n3(points), _f64(plane_point, (3,)), _f64(normal, (3,))
        out = np.empty_like(p)
        self._check(self._lib.f3d_points_plane_projection(self._h, _ptr(p), len(p), _ptr(pp), _ptr(nr), _ptr(out)))
        return out

    def lines_plane_projection(self, starts, ends, plane_point, normal):
        s, e, pp, nr = self._n3(starts), self._n3(ends), _f64(plane_point, (3,)), _f64(normal, (3,))
        sp, ep, dr = np.empty_like(s), np.empty_like(s), np.empty_like(s)
        self._check(self._lib.f3d_lines_plane_projection(self._h, _ptr(s), _ptr(e), len(s), _ptr(pp), _ptr(nr), _ptr(sp), _ptr(ep), _ptr(dr)))
        return sp, ep, dr

    def components_same_class(self, classes, offsets, neighbours):
        """root[i] = smallest index of point i's same-class connected component (CSR adjacency, symmetric)."""
        cls = np.ascontiguousarray(classes, dtype=np.int64)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        nb = np.ascontiguousarray(neighbours, dtype=np.int32)
        if len(offs) != len(cls) + 1 or (len(cls) and offs[-1] != len(nb)):
            raise ValueError('offsets must have n+1 entries ending at len(neighbours)')
        root = np.empty(len(cls), np.int64)
        self._check(self._lib.f3d_components_same_class(self._h, _ptr(cls), len(cls), _ptr(offs), _ptr(nb), _ptr(root)))
        return root

    def patch_owner(self, uv, seed_pts, seed_normals, frame_pts, frame_normals, free, h, w, half, radius, min_cosine):
        """owner int32 [h*w]: for every free depth pixel the first seed of Fusion.fuse's matching loop (fusion.py:269-298)
        that would take it, -1 if none."""
        uv = np.ascontiguousarray(uv, dtype=np.int32)
        sp, sn = _f64(seed_pts), _f64(seed_normals)
        qp, qn = _f64(frame_pts, (h * w, 3)), _f64(frame_normals, (h * w, 3))
        fr = np.ascontiguousarray(free, dtype=np.uint8).reshape(-1)
        m = len(sp)
        if uv.shape != (2, m) or sn.shape != (m, 3) or len(fr) != h * w:
            raise ValueError('patch_owner: uv must be [2,m], seeds [m,3], free [h*w]')
        owner = np.empty(h * w, np.int32)
        self._check(self._lib.f3d_patch_owner(self._h, _ptr(uv), m, h, w, int(half), float(radius), float(min_cosine), _ptr(sp), _ptr(sn),
                                              _ptr(qp), _ptr(qn), _ptr(fr), _ptr(owner)))
        return owner

    def patch_seeds(self, frame_pts, frame_normals, prio, free, h, w, half, radius, min_cosine):
        """Fusion.patch_downsample's seeds and what they take: owner int32 [h*w] (seed pixel index, -1 = nobody), rounds."""
        qp, qn = _f64(frame_pts, (h * w, 3)), _f64(frame_normals, (h * w, 3))
        pr = np.ascontiguousarray(prio, dtype=np.int32).reshape(-1)
        fr = np.ascontiguousarray(free, dtype=np.uint8).reshape(-1)
        if len(pr) != h * w or len(fr) != h * w:
            raise ValueError('patch_seeds: prio and free must have h*w entries')
        owner = np.empty(h * w, np.int32)
        rounds = C.c_int32(0)
        self._check(self._lib.f3d_patch_seeds(self._h, _ptr(qp), _ptr(qn), _ptr(pr), _ptr(fr), h, w, int(half), float(radius), float(min_cosine),
                                              _ptr(owner), C.byref(rounds)))
        return owner, rounds.value

    def patch_match(self, uv, seed_pts, seed_normals, frame_pts, frame_normals, frame_colors, free, h, w, half, radius, min_cosine):
        """patch_owner plus, per seed, the ordered sums of the frame rows it takes: (owner int32 [h*w], sums float64 [m, 9] =
        points | normals | colours, counts int32 [m])."""
        uv = np.ascontiguousarray(uv, dtype=np.int32)
        sp, sn = _f64(seed_pts), _f64(seed_normals)
        qp, qn = _f64(frame_pts, (h * w, 3)), _f64(frame_normals, (h * w, 3))
        qc = None if frame_colors is None else _f64(frame_colors, (h * w, 3))
        fr = np.ascontiguousarray(free, dtype=np.uint8).reshape(-1)
        m = len(sp)
        if uv.shape != (2, m) or sn.shape != (m, 3) or len(fr) != h * w:
            raise ValueError('patch_match: uv must be [2,m], seeds [m,3], free [h*w]')
        owner, sums, counts = np.empty(h * w, np.int32), np.zeros((m, 9)), np.zeros(m, np.int32)
        self._check(self._lib.f3d_patch_match(self._h, _ptr(uv), m, h, w, int(half), float(radius), float(min_cosine), _ptr(sp), _ptr(sn),
                                              _ptr(qp), _ptr(qn), _ptr(qc), _ptr(fr), _ptr(owner), _ptr(sums), _ptr(counts)))
        return owner, sums, counts

    def patch_seeds_sums(self, frame_pts, frame_normals, frame_colors, prio, free, h, w, half, radius, min_cosine):
        """patch_seeds plus the ordered sums per seed pixel: (owner int32 [h*w], sums float64 [h*w, 9], counts int32 [h*w], rounds)."""
        qp, qn = _f64(frame_pts, (h * w, 3)), _f64(frame_normals, (h * w, 3))
        qc = None if frame_colors is None else _f64(frame_colors, (h * w, 3))
        pr = np.ascontiguousarray(prio, dtype=np.int32).reshape(-1)
        fr = np.ascontiguousarray(free, dtype=np.uint8).reshape(-1)
        if len(pr) != h * w or len(fr) != h * w:
            raise ValueError('patch_seeds_sums: prio and free must have h*w entries')
        owner, sums, counts = np.empty(h * w, np.int32), np.zeros((h * w, 9)), np.zeros(h * w, np.int32)
        rounds = C.c_int32(0)
        self._check(self._lib.f3d_patch_seeds_sums(self._h, _ptr(qp), _ptr(qn), _ptr(qc), _ptr(pr), _ptr(fr), h, w, int(half), float(radius),
                                                   float(min_cosine), _ptr(owner), _ptr(sums), _ptr(counts), C.byref(rounds)))
        return owner, sums, counts, rounds.value

    def unproject_depth(self, depth, K, q_wxyz, t, depth_scale=1000.0):
        """Depth frame [H,W] (uint16, float32 or float64) -> world points float64 [H*W,3] (ios_rtab.py:171-173,187-192)."""
        d = np.ascontiguousarray(depth)
        if d.ndim != 2:
            raise ValueError('depth must be [H,W]')
        if d.dtype == np.uint16:
            code = 2
        elif d.dtype == np.float32:
            code = 1
        else:
            d, code = np.ascontiguousarray(d, dtype=np.float64), 0
        K, q, t = _f64(K, (3, 3)), _f64(q_wxyz, (4,)), _f64(t, (3,))
        out = np.empty((d.size, 3), np.float64)
        self._check(self._lib.f3d_unproject_depth(self._h, _ptr(d), code, d.shape[0], d.shape[1], _ptr(K), float(depth_scale), _ptr(q), _ptr(t), _ptr(out)))
        return out

    def radius_graph(self, points, radius):
        """KDTree(points).query_radius(points, r=radius) (fusion.py:374-375) as CSR: (offsets int64 [n+1], neighbours int32)."""
        p, dt = _xyz(points)
        offs = np.zeros(len(p) + 1, np.int64)
        nnz = C.c_int64(0)
        self._check(self._lib.f3d_radius_graph_count(self._h, _ptr(p), dt, len(p), float(radius), _ptr(offs), C.byref(nnz)))
        nb = np.empty(nnz.value, np.int32)
        self._check(self._lib.f3d_radius_graph_fill(self._h, len(p), _ptr(nb)))
        return offs, nb

    # ---------------------------------------------------------------- device-pointer calls
    def project_vote_argmax_dev(self, xyz_ptr, dtype, n, views_ptr, nviews, masks_ptr, h, w, nclasses, threshold,
                                filter_classes, classes_ptr, votes_ptr=None, stream=None, flags=0, perm_ptr=None):
        f, nf = _filter(filter_classes)
        self._check(self._lib.f3d_project_vote_argmax_dev(self._h, xyz_ptr, dtype, n, views_ptr, nviews, masks_ptr, h, w,
                                                          int(nclasses), _ptr(f), nf, float(threshold), classes_ptr,
                                                          votes_ptr, int(flags), perm_ptr, stream))

    # the fused path with the views arriving in chunks (f3d.h: f3d_mask_presence_dev .. f3d_fuse_chunk_dev)
    def mask_presence_dev(self, masks_ptr, nviews, h, w, present256_ptr, stream=None):
        self._check(self._lib.f3d_mask_presence_dev(self._h, masks_ptr, nviews, h, w, present256_ptr, stream))

    def fuse_chunked_begin_dev(self, present256_ptr, n, nviews, h, w, nclasses, filter_classes, stream=None):
        f, nf = _filter(filter_classes)
        self._check(self._lib.f3d_fuse_chunked_begin_dev(self._h, present256_ptr, n, nviews, h, w, int(nclasses), _ptr(f), nf, stream))

    def fuse_chunk_dev(self, xyz_ptr, dtype, n, views_ptr, nviews, v_begin, v_end, masks_ptr, h, w, nclasses, threshold,
                       filter_classes, classes_ptr, stream=None, flags=0, perm_ptr=None):
        f, nf = _filter(filter_classes)
        self._check(self._lib.f3d_fuse_chunk_dev(self._h, xyz_ptr, dtype, n, views_ptr, nviews, int(v_begin), int(v_end), masks_ptr, h, w,
                                                 int(nclasses), _ptr(f), nf, float(threshold), classes_ptr, int(flags), perm_ptr, stream))

    def coded_plane_bytes(self, h, w):
        return int(self._lib.f3d_coded_plane_bytes(int(h), int(w)))

    def code_planes_dev(self, masks_ptr, nplanes, h, w, coded_ptr, stream=None):
        self._check(self._lib.f3d_code_planes_dev(self._h, masks_ptr, int(nplanes), int(h), int(w), coded_ptr, stream))

    def fuse_chunk_coded_dev(self, xyz_ptr, dtype, n, views_ptr, nviews, v_begin, v_end, coded_ptr, h, w, nclasses, threshold,
                             filter_classes, classes_ptr, stream=None, flags=0, perm_ptr=None):
        f, nf = _filter(filter_classes)
        self._check(self._lib.f3d_fuse_chunk_coded_dev(self._h, xyz_ptr, dtype, n, views_ptr, nviews, int(v_begin), int(v_end), coded_ptr, h, w,
                                                       int(nclasses), _ptr(f), nf, float(threshold), classes_ptr, int(flags), perm_ptr, stream))

    def rotate_dev(self, xyz_ptr, n, q_wxyz, out_ptr, stream=None):
        q = _f64(q_wxyz, (4,))
        self._check(self._lib.f3d_rotate_f64_dev(self._h, xyz_ptr, n, _ptr(q), out_ptr, stream))

    def cloud_sort_cells_dev(self, xyz_ptr, dtype, n, sorted_ptr, perm_ptr, stream=None):
        self._check(self._lib.f3d_cloud_sort_cells_dev(self._h, xyz_ptr, dtype, n, sorted_ptr, perm_ptr, stream))

    def fastpath_audit(self, points, views, w=1024, h=1024):
        """(pairs inside, pairs left to the exact kernel, decided-but-different pairs, contradicted culls) on the cell-sorted cloud."""
        p, dt = _xyz(points)
        v = _f64(views)
        stats = np.zeros(4, np.uint64)
        self._check(self._lib.f3d_debug_fastpath_audit(self._h, _ptr(p), dt, len(p), _ptr(v), len(v), int(w), int(h), _ptr(stats)))
        return tuple(int(x) for x in stats)

    def fuse_deferred(self, stream=None):
        """(points the float32 kernel deferred to the float64 tier, points of those that needed the reference's own arithmetic) of the last fused call."""
        c = np.zeros(2, np.uint32)
        self._check(self._lib.f3d_debug_fuse_deferred(self._h, stream, _ptr(c)))
        return int(c[0]), int(c[1])

    def take_device_error(self, stream=None):
        self._check(self._lib.f3d_take_device_error(self._h, stream))

    def project_view_dev(self, xyz_ptr, dtype, n, view, uv_ptr, inside_ptr, stream=None):
        v = _f64(view, (VIEW_DOUBLES,))
        self._check(self._lib.f3d_project_view_dev(self._h, xyz_ptr, dtype, n, _ptr(v), uv_ptr, inside_ptr, stream))

    def segment_votes_dev(self, votes_ptr, npts, ncols, nclasses, threshold, filter_classes, classes_ptr, stream=None):
        f, nf = _filter(filter_classes)
        self._check(self._lib.f3d_segment_votes_dev(self._h, votes_ptr, npts, ncols, int(nclasses), float(threshold),
                                                    _ptr(f), nf, classes_ptr, stream))

    def vote_uv2pt_dev(self, uv2pt_ptr, mask_ptr, hw, votes_ptr, npts, ncols, stream=None):
        self._check(self._lib.f3d_vote_uv2pt_dev(self._h, uv2pt_ptr, mask_ptr, hw, votes_ptr, npts, ncols, stream))

    def vote_uv2pt_batch_dev(self, luts_ptr, masks_ptr, nframes, h, w, votes_ptr, npts, ncols, stream=None):
        self._check(self._lib.f3d_vote_uv2pt_batch_dev(self._h, luts_ptr, masks_ptr, int(nframes), int(h), int(w), votes_ptr, npts, ncols, stream))

    def sem_logits_to_mask_dev(self, sem_ptr, c, hw, conf, low_label, mask_ptr, stream=None):
        self._check(self._lib.f3d_sem_logits_to_mask_dev(self._h, sem_ptr, c, hw, float(conf or 0.0), int(low_label), mask_ptr, stream))

    def sem_logits_to_masks_dev(self, sem_ptr, nimg, c, hw, conf, low_label, masks_ptr, stream=None):
        """nimg images of logits [nimg, c, hw] -> nimg consecutive planes at masks_ptr (device-resident hand-off, no sync)."""
        self._check(self._lib.f3d_sem_logits_to_masks_dev(self._h, sem_ptr, int(nimg), c, hw, float(conf or 0.0), int(low_label), masks_ptr, stream))

    def points_in_obb_dev(self, xyz_ptr, dtype, n, boxes, bits_ptr, cooc_ptr, stream=None):
        b = _f64(boxes)
        self._check(self._lib.f3d_points_in_obb_dev(self._h, xyz_ptr, dtype, n, _ptr(b), len(b), bits_ptr, cooc_ptr, stream))

    def group_by_id_dev(self, ids_ptr, n, nids, order_ptr, sorted_ids_ptr, starts_ptr, stream=None):
        self._check(self._lib.f3d_group_by_id_dev(self._h, ids_ptr, n, int(nids), order_ptr, sorted_ids_ptr, starts_ptr, stream))

    def obb_extremes_dev(self, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, nids, extremes_ptr, stream=None):
        self._check(self._lib.f3d_obb_extremes_dev(self._h, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, int(nids), extremes_ptr, stream))

    def obb_hull_filter_dev(self, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, starts_ptr, nids, fstart_ptr, facets_ptr, margin_ptr,
                            cand_ptr, cand_count_ptr, stream=None):
        self._check(self._lib.f3d_obb_hull_filter_dev(self._h, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, starts_ptr, int(nids), fstart_ptr,
                                                      facets_ptr, margin_ptr, cand_ptr, cand_count_ptr, stream))

    def obb_candidates_dev(self, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, starts_ptr, nids, min_members, cand_ptr, cand_start_ptr, stream=None):
        self._check(self._lib.f3d_obb_candidates_dev(self._h, xyz_ptr, dtype, n, order_ptr, sorted_ids_ptr, starts_ptr, int(nids), int(min_members),
                                                     cand_ptr, cand_start_ptr, stream))

    def gather_points_dev(self, xyz_ptr, dtype, idx_ptr, count, out_ptr, stream=None):
        self._check(self._lib.f3d_gather_points_dev(self._h, xyz_ptr, dtype, idx_ptr, int(count), out_ptr, stream))

    def obb_fit_dev(self, pts_ptr, start_ptr, nfit, total, boxes_ptr, status_ptr, isvert_ptr, nvert_ptr=None, stream=None):
        self._check(self._lib.f3d_obb_fit_dev(self._h, pts_ptr, start_ptr, int(nfit), int(total), boxes_ptr, status_ptr, isvert_ptr, nvert_ptr, stream))

    def unproject_depth_batch_dev(self, depth_ptr, depth_code, nframes, h, w, K, q_wxyz, t, out_ptr, depth_scale=1000.0, stream=None):
        """F depth frames [F,h,w] on the device -> world points float64 [F,h*w,3] in one launch; q_wxyz [F,4], t [F,3] host arrays."""
        K, q, t = _f64(K, (3, 3)), _f64(q_wxyz, (int(nframes), 4)), _f64(t, (int(nframes), 3))
        self._check(self._lib.f3d_unproject_depth_batch_dev(self._h, depth_ptr, int(depth_code), int(nframes), int(h), int(w), _ptr(K), float(depth_scale),
                                                            _ptr(q), _ptr(t), out_ptr, stream))

    def relabel_dev(self, ids_ptr, n, from_id, to_id, count_ptr=None, stream=None):
        self._check(self._lib.f3d_relabel_dev(self._h, ids_ptr, n, int(from_id), int(to_id), count_ptr, stream))


_default = {}


def default_context(device=None):
    """Process-wide context per device (LOCAL_RANK selects the device under torchrun)."""
    if device is None:
        device = int(os.environ.get('F3D_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    ctx = _default.get(device)
    if ctx is None:
        ctx = _default[device] = Context(device)
    return ctx
