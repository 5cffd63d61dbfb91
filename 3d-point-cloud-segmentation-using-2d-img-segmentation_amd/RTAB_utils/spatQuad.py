"""Drop-in for the reference's RTAB_utils/spatQuad.py: quaternion helpers whose batched rotation runs
as a HIP kernel (f3d_rotate_f64).

pyquaternion is not a dependency here: the three members the pipeline uses -- the 4-sequence
constructor, ``.elements`` and ``.inverse`` (conjugate over the squared norm, NOT normalised;
camera_utils.py:22 relies on exactly that) -- are provided by this class, together with the
``axis=/angle=`` constructor getQuaternion needs (spatQuad.py:51).
"""
import numpy as np

import f3d


class SpatQuadranion:
    def __init__(self, *args, axis=None, angle=None):
        if axis is not None:
            axis = np.asarray(axis, np.float64)
            n = np.linalg.norm(axis)
            if n == 0:
                raise ZeroDivisionError('Provided rotation axis has no length')
            half = float(angle) / 2.0
            self.q = np.concatenate([[np.cos(half)], np.sin(half) * axis / n])
        elif len(args) == 1:
            self.q = np.array(args[0].q if isinstance(args[0], SpatQuadranion) else args[0], dtype=np.float64).reshape(4)
        elif len(args) == 4:
            self.q = np.array(args, dtype=np.float64)
        elif not args:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])
        else:
            raise ValueError('expected a (w, x, y, z) sequence')

    elements = property(lambda self: self.q)
    w = property(lambda self: self.q[0])
    x = property(lambda self: self.q[1])
    y = property(lambda self: self.q[2])
    z = property(lambda self: self.q[3])

    @property
    def inverse(self):
        return SpatQuadranion(f3d.quat_inverse(self.q))          # ZeroDivisionError for the zero quaternion

    def __mul__(self, other):                                   # Hamilton product (multiplyQuadernion, spatQuad.py:41-42)
        a, b = self.q, SpatQuadranion(other).q
        return SpatQuadranion([a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
                               a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                               a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
                               a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])

    def rotate(self, p):
        """[N,3] -> [N,3], q p q* without normalising q (reference spatQuad.py:7-28), on the GPU."""
        return f3d.default_context().rotate(np.asarray(p, np.float64).reshape(-1, 3), self.q)

    def __repr__(self):
        return f'TooliqaQuaternion{(self.w, self.x, self.y, self.z)}'

    __str__ = __repr__


def getQuaternion(v1, v2):
    """Rotation taking direction v1 to v2 (reference spatQuad.py:36-48)."""
    a = np.asarray(v1, np.float64) / np.linalg.norm(v1)
    b = np.asarray(v2, np.float64) / np.linalg.norm(v2)
    axis = np.cross(a, b)
    return SpatQuadranion(axis=axis / np.linalg.norm(axis), angle=np.arccos(np.dot(a, b)))


def multiplyQuadernion(q1, q2):
    return q2 * q1


def get_quaternion_from_euler(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return SpatQuadranion([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                           cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy])


def axis_transformation(points):
    return np.array(points)[:, :3].copy()
