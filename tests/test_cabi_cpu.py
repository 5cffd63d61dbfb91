"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/f3d.h
declares, the host-side view geometry equals the oracle, error mapping, and the product never touches oracle/."""
import re

import numpy as np
import pytest

import f3d
from conftest import ROOT, PKG, GOLDEN
from oracle import np_ref as O


def _header_functions():
    text = (ROOT / 'include' / 'f3d.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(f3d_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = f3d.library()
    declared = _header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} is declared in include/f3d.h but not exported by libf3d_hip.so'
    assert set(lib._f3d_symbols) == set(declared), set(lib._f3d_symbols) ^ set(declared)
    assert lib.f3d_version() == 100


def test_view_record_layout_matches_header():
    text = (ROOT / 'include' / 'f3d.h').read_text()
    body = text[text.index('typedef struct f3d_view {'):text.index('} f3d_view;')]
    doubles = sum(int(np.prod([int(x) if x.isdigit() else 5 for x in re.findall(r'\[(\w+)\]', m.group(2))] or [1]))
                  for m in re.finditer(r'\b(double)\s+\w+((?:\[\w+\])*);', body))
    floats = sum(int(np.prod([int(x) if x.isdigit() else 5 for x in re.findall(r'\[(\w+)\]', m.group(2))] or [1]))
                 for m in re.finditer(r'\b(float)\s+\w+((?:\[\w+\])*);', body))
    assert doubles * 8 + floats * 4 == f3d.VIEW_DOUBLES * 8 == 704


def test_views_build_equals_oracle_bit_for_bit(golden):
    g = golden('points2pixel')
    for K, (w, h) in zip(g['K'], [(720, 960), (512, 512), (1024, 1024)]):
        views = f3d.views_build(K, w, h, g['q_wxyz'], g['t'], 7.5)
        F = f3d.view_fields(views)
        pp, pn = O.frustum_planes(K, w, h, g['q_wxyz'], g['t'], 7.5)
        assert np.array_equal(F['plane_pt'], pp) and np.array_equal(F['plane_n'], pn)
        assert np.array_equal(F['K'], np.broadcast_to(K, F['K'].shape))
        assert np.array_equal(F['t'], g['t'])
        for j, q in enumerate(g['q_wxyz']):
            assert np.array_equal(F['qinv'][j], O.quat_inverse(q))
        eyes, look, nrm = f3d.frustum_data(K, w, h, g['q_wxyz'], g['t'])
        e2, l2, _, n2 = O.frustum_data(K, w, h, g['q_wxyz'], g['t'])
        assert np.array_equal(eyes, e2) and np.array_equal(look, l2) and np.array_equal(nrm, n2)


def test_frustum_data_matches_reference_golden(golden):
    g = golden('frustum')
    w, h = g['calib_wh']
    eyes, look, nrm = f3d.frustum_data(g['calib_K'], int(w), int(h), g['q_wxyz'], g['t'])
    assert np.abs(eyes - g['calib_eyes']).max() <= 1e-12
    assert np.abs(look - g['calib_lookats']).max() <= 1e-12
    assert np.abs(nrm - g['calib_face_normals']).max() <= 1e-12


def test_zero_quaternion_raises_like_pyquaternion():
    with pytest.raises(ZeroDivisionError):
        f3d.quat_inverse([0, 0, 0, 0])
    with pytest.raises(ZeroDivisionError):
        f3d.views_build(np.eye(3), 4, 4, [[0, 0, 0, 0]], [[0, 0, 0]], 1.0)
    assert np.array_equal(f3d.quat_inverse([2, 0, 0, 0]), [0.5, 0, 0, 0])          # Q4: conj / |q|^2


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    with pytest.raises(f3d.F3DUnavailable, match='no CPU fallback'):
        f3d.Context(0)
    from Fusion3DSeg.camera_utils import points2pixel
    with pytest.raises(f3d.F3DUnavailable):
        points2pixel(np.zeros((4, 3)), np.eye(3), [1, 0, 0, 0], [0, 0, 0])


def test_product_never_imports_the_oracle():
    offenders = []
    for path in list(PKG.rglob('*.py')) + list(PKG.rglob('*.hip')) + list(PKG.rglob('*.cpp')) + list(PKG.rglob('*.h')):
        text = path.read_text()
        code = re.sub(r'//.*|#(?!include).*', '', re.sub(r'/\*.*?\*/|""".*?"""', '', text, flags=re.S))   # comments may cite it
        if re.search(r'^\s*(from|import)\s+oracle\b', code, re.M) or re.search(r'#include\s*[<"][^>"]*oracle', code) \
                or 'libf3d_oracle' in code or 'c_ref' in code or re.search(r'\bnp_ref\b', code):
            offenders.append(str(path))
    assert not offenders, offenders


def test_synthetic_scene_is_deterministic():
    from f3d import synth
    a, b = synth.scene('C1', n=1000), synth.scene('C1', n=1000)
    assert np.array_equal(a['points'], b['points']) and np.array_equal(a['masks'], b['masks'])
    assert np.array_equal(a['points'], a['points'].astype(np.float32).astype(np.float64))   # f32-representable
    assert set(np.unique(a['masks'])) <= set(synth.ALPHABET.tolist())
    assert np.allclose(np.linalg.norm(a['wxyzs'], axis=1), 1.0)
