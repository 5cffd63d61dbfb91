#!/usr/bin/env python3
"""Golden vectors for the mask post-processing of get2DSeg.SegmentImage.

Run in the build container only (needs /root/reference and CPU torch):
    python tests/golden/make_golden_sem.py

The post-processing is not a function of its own in the reference: it is six statements inside the per-image
loop of ``SegmentImage`` (get2DSeg.py:111-120): ``sem.argmax(dim=0)``, ``T.nn.Softmax(dim=0)``, ``T.amax``,
``< conf_threshold``, the assignment of 133 and the conversion to NumPy.  This script takes exactly those
statements out of the reference's syntax tree at generation time (nothing is imported from get2DSeg.py, whose
header needs detectron2 / OneFormer / cv2, absent from the image), wraps them in a function whose arguments are
the names they read (``sem``, ``conf_threshold``) and runs them unmodified with CPU torch on seeded logits.
The network that produces ``sem`` is third-party and absent; only its output's post-processing is pinned.

Cases (float32 logits [C, H, W]):
* ``img``    C = 133, random logits with a block of nearly flat pixels (max probability ~ 1/133 < 0.017 -> 133);
* ``edge``   C = 133, one raised logit per pixel placed so that the maximum probability sits at relative offsets
             of 3e-5 ... 1e-2 on either side of conf_threshold = 0.017 (plus small noise on the other logits);
* ``tie``    C = 133 and C = 5, equal maxima (first maximum must win), all-equal logits, and conf_threshold = 0
             (the ``if conf_threshold:`` branch not taken);
* ``small``  C = 8, threshold 0.3, straddling pixels, a width that is not a multiple of 4.

Stored per case: the logits, conf_threshold, the reference's uint8/int64 mask, and torch's own float32 maximum
probability (so that the tests can state the band around the threshold they do not compare in).
"""
import ast
import sys
from pathlib import Path

import numpy as np
import torch as T

REF = Path('/root/reference')
OUT = Path(__file__).resolve().parent
sys.dont_write_bytecode = True


def reference_postprocess():
    """The statements of get2DSeg.py:111-120 as ``f(sem, conf_threshold) -> np.ndarray``, compiled from the reference."""
    tree = ast.parse((REF / 'get2DSeg.py').read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'SegmentImage')
    loop = next(n for n in fn.body if isinstance(n, ast.For))
    keep = []
    for st in loop.body:
        src = ast.unparse(st)
        if isinstance(st, ast.Assign) and src.startswith('sem_image = sem.argmax'):
            keep.append(st)                                   # :111
        elif isinstance(st, ast.If) and src.startswith('if conf_threshold'):
            keep.append(st)                                   # :113-118
        elif isinstance(st, ast.Assign) and src.startswith('sem_image = np.array(sem_image.cpu())'):
            keep.append(st)                                   # :120
    assert len(keep) == 3, [ast.unparse(k) for k in keep]
    args = ast.arguments(posonlyargs=[], args=[ast.arg('sem'), ast.arg('conf_threshold')], kwonlyargs=[], kw_defaults=[], defaults=[])
    body = keep + [ast.Return(ast.Name('sem_image', ast.Load()))]
    f = ast.FunctionDef(name='postprocess', args=args, body=body, decorator_list=[], lineno=1, col_offset=0)
    mod = ast.fix_missing_locations(ast.Module(body=[f], type_ignores=[]))
    ns = {'T': T, 'np': np}
    exec(compile(mod, str(REF / 'get2DSeg.py'), 'exec'), ns)
    return ns['postprocess']


def pmax_of(sem):
    """torch's own float32 maximum softmax probability (what get2DSeg.py:114-116 compares with the threshold)."""
    return T.amax(T.nn.Softmax(dim=0)(T.from_numpy(sem)), dim=0).numpy()


def main():
    post = reference_postprocess()
    rng = np.random.default_rng(20241004)
    out = {}

    def case(name, sem, thr):
        sem = np.ascontiguousarray(sem, dtype=np.float32)
        out[f'{name}_sem'] = sem
        out[f'{name}_conf'] = np.array(thr, np.float64)
        out[f'{name}_mask'] = post(T.from_numpy(sem.copy()), thr)
        out[f'{name}_pmax'] = pmax_of(sem)

    # img: quantised to 1/64 so that the fixture compresses; a band of nearly flat pixels
    sem = np.round(rng.normal(size=(133, 24, 36)) * 3 * 64) / 64
    sem[:, :5, :] = np.round(sem[:, :5, :] * 0.01 * 1024) / 1024
    case('img', sem, 0.017)

    # edge: p = e^a / (e^a + sum_others) around 0.017
    C, n = 133, 24 * 32
    base = np.round(rng.normal(size=(C, n)) * 0.02 * 4096) / 4096
    rel = np.concatenate([s * np.array([3e-5, 1e-4, 3e-4, 1e-3, 1e-2]) for s in (1, -1)])
    for i in range(n):
        k = rng.integers(0, C)
        others = np.exp(np.delete(base[:, i], k).astype(np.float64)).sum()
        p = 0.017 * (1 + rel[i % len(rel)])
        base[k, i] = np.log(p * others / (1 - p))
    case('edge', base.reshape(C, 24, 32), 0.017)

    # tie: equal maxima, all-equal logits; with and without the threshold branch
    t = np.round(rng.normal(size=(133, 6, 8)) * 64) / 64
    t[40, 0, :] = 9.0; t[7, 0, :] = 9.0; t[100, 0, :] = 9.0          # three equal maxima: index 7 wins
    t[:, 1, :] = 0.25                                                # all equal: index 0, p = 1/133 -> 133 with the threshold
    t[132, 2, :] = 11.0; t[0, 2, :] = 11.0                           # first and last
    case('tie', t, 0.017)
    case('tie0', t, 0)                                               # conf_threshold falsy: plain argmax
    t5 = np.zeros((5, 3, 7), np.float32)
    t5[3, 1, :] = 1.0; t5[1, 1, :] = 1.0
    case('tie5', t5, 0.1)

    # small: C = 8, threshold 0.3, odd width
    C, h, w = 8, 20, 37
    s8 = np.round(rng.normal(size=(C, h * w)) * 0.5 * 1024) / 1024
    for i in range(0, h * w, 3):
        k = rng.integers(0, C)
        others = np.exp(np.delete(s8[:, i], k).astype(np.float64)).sum()
        p = 0.3 * (1 + rel[(i // 3) % len(rel)])
        s8[k, i] = np.log(p * others / (1 - p))
    case('small', s8.reshape(C, h, w), 0.3)

    np.savez_compressed(OUT / 'sem_mask.npz', **out)
    print(f"sem_mask.npz {(OUT / 'sem_mask.npz').stat().st_size / 1024:.1f} KiB; "
          + ', '.join(f"{k[:-5]}: {int((out[k] == 133).sum())} of {out[k].size} -> 133" for k in out if k.endswith('_mask')))


if __name__ == '__main__':
    main()
