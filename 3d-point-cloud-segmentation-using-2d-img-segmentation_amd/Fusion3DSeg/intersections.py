"""Drop-in for the reference's Fusion3DSeg/intersections.py.

``point_inside_polyhedra`` (reference :146-164) is the hot function -- it runs on every fused point per frame
(fusion.py:260) -- and is a HIP kernel.  The other batched primitives of the file, which no entry point of the
reference calls, are HIP kernels too (one thread per output element, csrc/f3d_geom.hip); the two that act on a single
pair of vectors (``plane_x_plane``, ``ray_ray_closest``) are a handful of flops and stay on the host.
"""
import numpy as np

import f3d


def _ctx():
    return f3d.default_context()


def point_inside_polyhedra(points, plane_points, normals):
    """bool [N]: point n is inside iff (p_n - plane_point_m) . normal_m >= 0 for every plane m."""
    return _ctx().inside_polyhedra(points, plane_points, normals)


def ray_x_lines(origin, direction, starts, ends):
    """Intersection of one ray with N coplanar segments -> (points [N,3], hit-inside-segment-and-ahead [N])."""
    return _ctx().ray_x_lines(origin, direction, starts, ends)


def rays_x_plane(plane_point, plane_normal, origins, directions):
    """N rays against one plane -> (points [N,3], valid [N]); only rays heading into the plane (denominator < -1e-6) are valid."""
    return _ctx().rays_x_plane(plane_point, plane_normal, origins, directions)


def lines_x_planes(line_origins, line_ends, plane_points, plane_normals):
    """N segments against M planes -> (points [N,M,3], valid [N,M]).  Like the reference (which broadcasts [N,3] against
    [N,M,3] without a new axis) this only accepts N == 1 or N == M and raises ValueError otherwise."""
    return _ctx().lines_x_planes(line_origins, line_ends, plane_points, plane_normals)


def point_inside_polygon(points, vertices):
    """-> (inside [N], within_boundary [M,N]) for a planar polygon given by M ordered vertices."""
    return _ctx().point_inside_polygon(points, vertices)


def plane_x_plane(n1=None, v1=None, n2=None, v2=None, lookat=None):
    """Unit direction of the line two planes share; a plane is given by its normal or by three of its points."""
    n1 = np.cross(v1[1] - v1[0], v1[2] - v1[0]) if n1 is None else n1
    n2 = np.cross(v2[1] - v2[0], v2[2] - v2[0]) if n2 is None else n2
    line = np.cross(n1, n2)
    line = line / np.linalg.norm(line)
    if lookat is not None and not line.dot(lookat) > 0:
        line = -line
    return line


def points_plane_projection(points, plane_point, normal):
    return _ctx().points_plane_projection(points, plane_point, normal)


def lines_plane_projection(starts, ends, plane_point, normal):
    """-> (start projections, end projections, unit directions of the projected segments), each [N,3]."""
    return _ctx().lines_plane_projection(starts, ends, plane_point, normal)


def ray_ray_closest(a0, a1, b0, b1):
    """Closest points of the lines through segments a0->a1 and b0->b1:
    (pa, pb, distance, intersects (< 1e-6), pa within segment a, pb within segment b)."""
    ua, ub = a1 - a0, b1 - b0
    la, lb = np.linalg.norm(ua), np.linalg.norm(ub)
    ua, ub = ua / la, ub / lb
    w = np.cross(ua, ub)
    den = np.linalg.norm(w) ** 2
    ab = b0 - a0
    with np.errstate(all='ignore'):
        ta = np.linalg.det(np.array([ab, ub, w])) / den
        tb = np.linalg.det(np.array([ab, ua, w])) / den
        pa, pb = a0 + ua * ta, b0 + ub * tb
        dist = np.linalg.norm(pa - pb)
        return pa, pb, dist, dist < 1e-6, np.linalg.norm(pa - a0) <= la, np.linalg.norm(pb - b0) <= lb
