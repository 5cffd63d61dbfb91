#!/usr/bin/env python3
"""Golden vectors for row (f)#3: RTAB2Cache.__getModP3d (RTAB_utils/ios_rtab.py:179-193), run from the reference.

Run in the build container (the reference is mounted at /root/reference): ``python tests/golden/make_golden_rtab.py``.
Like make_golden.py, the class definition is compiled from the reference's file by ``ast`` (the module itself imports PIL,
cv2 and skimage readers that the method does not touch); nothing of it is copied.  ``SpatQuadranion`` is the reference's
class over the same three-member restatement of pyquaternion (parity unpinned for that constructor; here it also has
to accept the four ``str(float)`` arguments ios_rtab.py:190 passes, which pyquaternion converts with ``float()``).

``__getRGBP3d`` (:155-177) cannot be run: it calls skimage's ``resize`` for the colours.  Its three depth lines (:171-173)
are restated in oracle/np_ref.py::unproject_depth; the ``orig_ptx`` fed to the reference here is produced by that
restatement, so the fixture pins everything downstream of it: the /1000, the quaternion reorder + string round trip, the
rotation and the translation.
"""
import ast
import sys
from pathlib import Path

import numpy as np

REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parent.parent))
sys.path.insert(0, str(OUT))
from make_golden import _defs_from, _QuaternionBase  # noqa: E402
from oracle import np_ref as O  # noqa: E402


class _Quaternion4(_QuaternionBase):
    def __init__(self, *args):
        super().__init__([float(a) for a in (args[0] if len(args) == 1 else args)])


def main():
    ns_q = {'np': np, 'Quaternion': _Quaternion4}
    _defs_from('RTAB_utils/spatQuad.py', ['SpatQuadranion'], ns_q)
    ns = {'np': np, 'SpatQuadranion': ns_q['SpatQuadranion']}
    _defs_from('RTAB_utils/ios_rtab.py', ['RTAB2Cache'], ns)
    cache = object.__new__(ns['RTAB2Cache'])                          # no files: only the attributes the method reads
    rng = np.random.default_rng(20240612)
    K = np.array([[798.94403076171875 * 256 / 1440, 0., 361.95578002929688 * 256 / 1440],
                  [0., 798.94403076171875 * 192 / 1920, 474.56329345703125 * 192 / 1920],
                  [0., 0., 1.]])
    F, H, W = 3, 24, 32
    depths = rng.integers(0, 6000, (F, H, W)).astype(np.uint16)
    depths[:, 0, 0] = 0
    odo_xyzw = rng.normal(size=(F, 4))
    odo_xyzw[0] /= np.linalg.norm(odo_xyzw[0])                         # one unit pose, two un-normalised ones
    odo_xyz = rng.normal(size=(F, 3)) * 2
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    # camera-frame points in millimetres, as __getRGBP3d leaves them (restated lines :171-173; scale 1, identity pose)
    orig = [O.unproject_depth(d, K, ident, np.zeros(3), depth_scale=1) for d in depths]
    cache.orig_ptx = orig
    cache.odo_wxyz = odo_xyzw                                          # the reference's name for the (x, y, z, w) columns
    cache.odo_xyz = odo_xyz
    mod = cache._RTAB2Cache__getModP3d()
    np.savez_compressed(OUT / 'modp3d.npz', K=K, depths=depths, odo_xyzw=odo_xyzw, odo_xyz=odo_xyz,
                        orig_ptx=np.stack(orig), mod_ptx=np.stack(mod))
    print('modp3d.npz', (OUT / 'modp3d.npz').stat().st_size, 'bytes')


if __name__ == '__main__':
    main()
