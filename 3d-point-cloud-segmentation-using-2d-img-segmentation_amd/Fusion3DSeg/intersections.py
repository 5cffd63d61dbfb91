"""Drop-in for the hot function of the reference's Fusion3DSeg/intersections.py.

``point_inside_polyhedra`` (reference :146-164, called on every fused point per frame at
fusion.py:260) runs as a HIP kernel.  The other nine primitives of that file are not called by any
entry point of the reference and are scheduled after the hot path.
"""
import f3d


def point_inside_polyhedra(points, plane_points, normals):
    """bool [N]: point n is inside iff (p_n - plane_point_m) . normal_m >= 0 for every plane m."""
    return f3d.default_context().inside_polyhedra(points, plane_points, normals)
