"""Hot-path slice of the reference's RTAB_utils/ios_rtab.py: depth frame -> world points on the GPU (row (f)#3).

``RTAB2Cache`` itself (pose-file / PNG / JPEG readers, colour resize, normals, the pickle cache) stays out of scope: it is
file I/O around the two private methods restated here, ``__getRGBP3d`` (:155-177) and ``__getModP3d`` (:179-193).
"""
import numpy as np

import f3d


def resize_camera_matrix(intrinsic, scale_x, scale_y):
    """RTAB2Cache.__resize_camera_matrix (:115-131): intrinsics of the depth resolution."""
    K = np.asarray(intrinsic, np.float64)
    return np.array([[K[0, 0] * scale_x, 0., K[0, 2] * scale_x],
                     [0., K[1, 1] * scale_y, K[1, 2] * scale_y],
                     [0., 0., 1.0]])


def frame_points_world(depth, intrinsics_scaled, odo_xyzw, odo_xyz, depth_scale=1000):
    """One frame of ``mod_ptx``: unproject ``depth`` [H,W] with the scaled intrinsics (:171-173), mm -> m (:187), rotate by
    the pose quaternion -- ``odo_xyzw`` is the pose file's (x, y, z, w) row, as the reference indexes it (:190) -- and add
    the translation (:191-192).  Returns float64 [H*W, 3] in row-major pixel order."""
    q = np.asarray(odo_xyzw, np.float64)
    return f3d.default_context().unproject_depth(depth, intrinsics_scaled, q[[3, 0, 1, 2]], odo_xyz, depth_scale)


def frames_points_world(depths, intrinsics_scaled, odo_xyzw, odo_xyz, depth_scale=1000):
    """``RTAB2Cache.__getModP3d`` over all frames: list of [H*W, 3] arrays."""
    return [frame_points_world(d, intrinsics_scaled, q, t, depth_scale) for d, q, t in zip(depths, odo_xyzw, odo_xyz)]
