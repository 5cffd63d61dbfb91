"""Seeded synthetic inputs of SURVEY.md 8(d): cloud (seed 1234), ring cameras (5678), masks (9012).

Data generation only (NumPy); shared by tests/ and bench.py so that the GPU path, the oracle and
the CPU baseline all see the same bytes.
"""
import numpy as np

CALIB_K = np.array([[798.94403076171875, 0., 361.95578002929688],
                    [0., 798.94403076171875, 474.56329345703125],
                    [0., 0., 1.]])                     # RTAB_utils/calibration.yaml:6-9 (720 x 960)
ALPHABET = np.array([0, 15, 86, 114, 115, 120, 132, 133], np.uint8)

CONFIGS = {                                             # BASELINE.json configs
    'C1': dict(n=100_000, views=4, w=512, h=512, K=np.array([[400., 0, 256], [0, 400., 256], [0, 0, 1]])),
    'C2': dict(n=1_000_000, views=16, w=720, h=960, K=CALIB_K),
    'C3': dict(n=10_000_000, views=64, w=1024, h=1024, K=np.array([[800., 0, 512], [0, 800., 512], [0, 0, 1]])),
    'C5': dict(n=50_000_000, views=256, w=1024, h=1024, K=np.array([[800., 0, 512], [0, 800., 512], [0, 0, 1]])),
}


def cloud(n, seed=1234, dtype=np.float64, shard=0):
    """n points uniform in x,y in [-5,5], z in [0,3], drawn as float32 (so f32 and f64 storage agree).
    `shard` > 0 selects an independent stream (the shard of one big cloud a rank owns)."""
    rng = np.random.default_rng(seed if shard == 0 else [seed, shard])
    lo = np.array([-5, -5, 0], np.float32)
    hi = np.array([5, 5, 3], np.float32)
    out = np.empty((n, 3), np.float32)
    step = 4_000_000
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s:s + m] = lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)
    return out if dtype == np.float32 else out.astype(np.float64)


def _look_at_quat(eye, target):
    z = target - eye
    z = z / np.linalg.norm(z)
    x = np.cross(z, np.array([0., 0., 1.]))
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)                                   # x right, y down, z forward
    R = np.stack([x, y, z], axis=1)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def ring_views(V, seed=5678):
    """Camera j at (4cos, 4sin, 1.5) looking at (0,0,1.5)+U(-.5,.5)^3; returns wxyz [V,4] (camera->world), t [V,3]."""
    rng = np.random.default_rng(seed)
    th = 2 * np.pi * np.arange(V) / V
    eyes = np.stack([4 * np.cos(th), 4 * np.sin(th), np.full(V, 1.5)], axis=1)
    targets = np.array([0., 0., 1.5]) + rng.uniform(-0.5, 0.5, (V, 3))
    quats = np.stack([_look_at_quat(e, t) for e, t in zip(eyes, targets)])
    return quats, eyes


def masks(V, h, w, kind='block64', seed=9012):
    """uint8 [V,h,w]: 'iid' = every pixel uniform over 0..133; 'block64' = 64x64 blocks from ALPHABET."""
    rng = np.random.default_rng(seed)
    if kind == 'iid':
        return rng.integers(0, 134, (V, h, w), dtype=np.uint8)
    if kind == 'block64x40':                              # 40 labels: the packed-bin instance of the fused kernel
        alphabet = np.arange(0, 133, 133 // 40, dtype=np.uint8)[:40]
    elif kind in ('block64x96', 'block64x100'):           # 96 / 100 labels: the large packed instance / the any-alphabet instance
        alphabet = np.arange(0, int(kind[8:]), dtype=np.uint8)
    elif kind == 'block64':
        alphabet = ALPHABET
    else:
        raise ValueError(kind)
    bh, bw = (h + 63) // 64, (w + 63) // 64
    blocks = alphabet[rng.integers(0, len(alphabet), (V, bh, bw))]
    return np.ascontiguousarray(np.repeat(np.repeat(blocks, 64, axis=1), 64, axis=2)[:, :h, :w])


def scene(name, n=None, mask_kind='block64', dtype=np.float64):
    cfg = CONFIGS[name]
    n = cfg['n'] if n is None else n
    q, t = ring_views(cfg['views'])
    return dict(points=cloud(n, dtype=dtype), K=cfg['K'], w=cfg['w'], h=cfg['h'], wxyzs=q, translations=t,
                masks=masks(cfg['views'], cfg['h'], cfg['w'], mask_kind), max_depth=10.0, nclasses=133)


def depth_sequence(h=192, w=256, nframes=6, focal=210.0, step=0.15, seed=7):
    """A capture for Fusion.fuse: a wall at z = 2.5 m and a floor seen by a camera that slides along x (axes = world axes).
    Returns K, wxyz [F,4], t [F,3] and per frame world points, normals, colours [h*w,3] and a validity mask [h*w]."""
    rng = np.random.default_rng(seed)
    K = np.array([[focal, 0.0, w / 2.0], [0.0, focal, h / 2.0], [0.0, 0.0, 1.0]])
    q = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (nframes, 1))
    t = np.stack([np.array([step * j, 0.0, 0.0]) for j in range(nframes)])
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    dirx, diry = (uu - K[0, 2]) / K[0, 0], (vv - K[1, 2]) / K[1, 1]
    frames = []
    for j in range(nframes):
        with np.errstate(divide='ignore'):
            z_floor = np.where(diry > 1e-9, 0.6 / diry, np.inf)
        z = np.minimum(2.5, z_floor) + rng.normal(0, 0.002, (h, w))
        cam = np.stack([dirx * z, diry * z, z], -1).reshape(-1, 3)
        nrm = np.where((z_floor < 2.5).reshape(-1, 1), np.array([0.0, -1.0, 0.0]), np.array([0.0, 0.0, -1.0]))
        valid = np.ones(h * w, bool)
        valid[rng.integers(0, h * w, h * w // 30)] = False
        frames.append((str(j), cam + t[j], nrm, rng.uniform(0, 1, (h * w, 3)), valid))
    return K, q, t, frames
