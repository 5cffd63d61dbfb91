"""Parity of the HIP path (through the C-ABI, via ctypes) against the oracle and the golden vectors.
Needs a real MI355X: run with `-m gpu`.  Integer outputs (labels, uv, inside, votes) are compared
bit for bit with the oracle on the same seeded inputs."""
import numpy as np
import pytest

import f3d
from f3d import synth
from oracle import np_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    return f3d.default_context()


def test_rotate_bit_exact(ctx, golden):
    g = golden('rotate')
    for q in g['q_wxyz']:
        assert np.array_equal(ctx.rotate(g['points'], q), O.rotate(q, g['points']))
    assert ctx.rotate(np.zeros((0, 3)), g['q_wxyz'][0]).shape == (0, 3)


def test_points2pixel_vs_oracle_and_golden(ctx, golden):
    g = golden('points2pixel')
    flips = total = 0
    for ki, K in enumerate(g['K']):
        for j, (q, t) in enumerate(zip(g['q_wxyz'], g['t'])):
            got = ctx.points2pixel(g['points'], K, q, t)
            assert got.dtype == np.int32 and got.shape == (2, len(g['points']))
            assert np.array_equal(got, O.points2pixel(g['points'], K, q, t))        # bit-exact vs oracle
            want = g['uv'][ki, j]
            ok = (np.abs(want.astype(np.int64)) < 2 ** 30).all(0)
            d = np.abs(got.astype(np.int64) - want)[:, ok]
            assert d.max() <= 1                                                     # stated uv tolerance vs reference
            flips += int((d != 0).sum()); total += d.size
    assert flips <= int(1e-6 * total)


def test_points2pixel_degenerate_inputs(ctx):
    K = synth.CALIB_K
    q, t = np.array([1., 0, 0, 0]), np.zeros(3)
    pts = np.array([[0., 0, 0], [1, 1, 0], [1e300, 1, 1e-300], [np.nan, 0, 1], [0.5, 0.25, 1.0], [-3., 2., -1.]])
    with np.errstate(all='ignore'):
        want = O.points2pixel(pts, K, q, t)
    assert np.array_equal(ctx.points2pixel(pts, K, q, t), want)                     # z = 0 / NaN -> INT32_MIN
    with pytest.raises(ZeroDivisionError):
        ctx.points2pixel(pts, K, np.zeros(4), t)


def test_inside_polyhedra_bit_exact_incl_near_plane(ctx, golden):
    g = golden('inside_polyhedra')
    for j in range(len(g['plane_points'])):
        got = ctx.inside_polyhedra(g['points'], g['plane_points'][j], g['plane_normals'][j])
        assert got.dtype == np.bool_ and np.array_equal(got, g['inside'][j])
        got = ctx.inside_polyhedra(g['adv_points'], g['plane_points'][j], g['plane_normals'][j])
        assert np.array_equal(got, g['inside_adv'][j])
    # more planes than one launch carries (chained launches), and the empty plane list
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (3000, 3))
    nr = rng.normal(size=(40, 3)); pp = -0.9 * nr / np.linalg.norm(nr, axis=1)[:, None]
    assert np.array_equal(ctx.inside_polyhedra(pts, pp, nr), O.point_inside_polyhedra(pts, pp, nr))
    assert ctx.inside_polyhedra(pts, np.zeros((0, 3)), np.zeros((0, 3))).all()


def test_project_view_single_view_fused(ctx):
    sc = synth.scene('C2', n=20000)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    ppts, pnrm = O.frustum_planes(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    for j in (0, 7):
        uv, ins = ctx.project_view(sc['points'], views[j])
        assert np.array_equal(uv, O.points2pixel(sc['points'], sc['K'], sc['wxyzs'][j], sc['translations'][j]))
        assert np.array_equal(ins, O.point_inside_polyhedra(sc['points'], ppts[j], pnrm[j]))
        assert 0.05 < ins.mean() < 0.95


@pytest.mark.parametrize('mask_kind', ['block64', 'iid'])
@pytest.mark.parametrize('thr,flt', [(0.5, None), (0.0, None), (0.5, [86, 114, 115]), (0.3, [115, 0, 86]),
                                     (0.2, [3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 120, 15, 133, 132, 86, 114])])
def test_fused_forward_labels_and_votes_bit_exact(ctx, mask_kind, thr, flt):
    sc = synth.scene('C1', n=30000, mask_kind=mask_kind)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    want_cls, want_votes = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'],
                                                 sc['max_depth'], 133, thr, flt, return_votes=True)
    got = ctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, thr, flt)
    assert got.dtype == np.int64 and np.array_equal(got, want_cls)
    got2, votes = ctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, thr, flt, return_votes=True)
    assert np.array_equal(got2, want_cls)
    assert np.array_equal(votes.astype(np.float64), want_votes)
    assert want_votes.sum() > 0.2 * len(sc['points'])
    # float32 storage of the same cloud gives the same labels (points are f32-representable)
    got32 = ctx.project_vote_argmax(sc['points'].astype(np.float32), views, sc['masks'], 133, thr, flt)
    assert np.array_equal(got32, want_cls)


def test_fused_forward_many_views_uint16_bins(ctx):
    rng = np.random.default_rng(3)
    V, h, w = 300, 64, 64
    K = np.array([[50., 0, 32], [0, 50., 32], [0, 0, 1]])
    q, t = synth.ring_views(V)
    pts = synth.cloud(5000)
    masks = np.full((V, h, w), 86, np.uint8)                      # every visible view votes 86: counts > 255
    masks[::7] = rng.integers(0, 134, (len(masks[::7]), h, w), dtype=np.uint8)
    views = f3d.views_build(K, w, h, q, t, 10.0)
    want, wv = O.project_vote_argmax(pts, K, q, t, masks, 10.0, 133, 0.5, None, return_votes=True)
    got, gv = ctx.project_vote_argmax(pts, views, masks, 133, 0.5, None, return_votes=True)
    assert wv.max() > 255
    assert np.array_equal(got, want) and np.array_equal(gv.astype(np.float64), wv)
    assert np.array_equal(ctx.project_vote_argmax(pts, views, masks, 133, 0.5, None), want)


@pytest.mark.parametrize('V', [240, 247, 255])
def test_fused_forward_view_counts_at_the_8bit_boundary(ctx, V):
    """A point casts more votes than there are views (placeholders of the software pipeline, padding slots): with packed 8-bit
    bins the "no sample" byte must not carry into the rejected-label bin.  iid masks (the packed instances),
    n not a multiple of the 128-point wave tile, a camera ring that leaves most points with few hits; one-shot and as ONE chunk of a
    chunked call (the CARRY instance), both against the oracle."""
    h = w = 96
    K = np.array([[70., 0, 48], [0, 70., 48], [0, 0, 1]])
    q, t = synth.ring_views(V)
    pts = synth.cloud(20_037, seed=V)
    rng = np.random.default_rng(V)
    views = f3d.views_build(K, w, h, q, t, 3.0)                              # short frusta: most (point, view) pairs miss
    for nlab in (134, 40):                                                   # any-alphabet and 48-code packed instances
        masks = rng.choice(134, nlab, replace=False).astype(np.uint8)[rng.integers(0, nlab, (V, h, w))]
        sub = rng.choice(len(pts), 2500, replace=False)
        want, wv = O.project_vote_argmax(pts[sub], K, q, t, masks, 3.0, 133, 0.0, None, return_votes=True)
        got = _dev_fuse(ctx, pts, views, masks, None, 0.0, f3d.FUSE_SORT)
        assert np.array_equal(got[sub], want)
        assert (wv.sum(1) <= 8).mean() > 0.05                                # points with few hits exist: their byte 0 is the one at risk
        one_chunk = _dev_fuse_chunked(ctx, pts, views, masks, None, 0.0, f3d.FUSE_SORT, [0, V])
        assert np.array_equal(one_chunk, got)
        two = _dev_fuse_chunked(ctx, pts, views, masks, None, 0.0, 0, [0, V - 3, V])
        assert np.array_equal(two, got)


def test_fused_forward_edge_cases(ctx):
    sc = synth.scene('C1', n=1000)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    assert ctx.project_vote_argmax(np.zeros((0, 3)), views, sc['masks']).shape == (0,)
    # points exactly on planes / at the eye: the exact plane test decides
    eye = sc['translations'][0]
    pts = np.vstack([eye, eye + 1e-13, sc['points'][:50], [[np.nan, 0, 0]], [[np.inf, 0, 0]], [[1e308, 1e308, 1e308]]])
    with np.errstate(all='ignore'):
        want = O.project_vote_argmax(pts, sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'])
    assert np.array_equal(ctx.project_vote_argmax(pts, views, sc['masks']), want)
    # a label beyond nclasses raises IndexError like voting.py:98
    bad = sc['masks'].copy(); bad[:] = 200
    with pytest.raises(IndexError):
        ctx.project_vote_argmax(sc['points'], views, bad, nclasses=133)
    with pytest.raises(IndexError):
        ctx.project_vote_argmax(sc['points'], views, sc['masks'], nclasses=133, filter_classes=[86, 134])


def test_vote_uv2pt_q1_and_segment_golden(ctx, golden):
    g = golden('voting')
    ncls = int(g['nclasses'])
    votes = np.zeros_like(g['votes'])
    for mask, lut in zip(g['masks'], g['uv2pt']):
        ctx.vote_uv2pt(votes, lut, mask.reshape(-1))
    assert np.array_equal(votes, g['votes'])
    for i in range(int(g['nseg'])):
        flt = g[f'seg{i}_filter'].tolist() if g[f'seg{i}_has_filter'] else None
        got = ctx.segment_votes(g['votes'], ncls, float(g[f'seg{i}_threshold']), flt)
        assert got.dtype == np.int64 and np.array_equal(got, g[f'seg{i}_classes']), i
    assert np.array_equal(ctx.segment_votes(g['votes'], g['votes'].shape[1], 0.75, None), g['segq2_classes'])   # Q2
    for i in range(int(g['nsmall'])):                                # odd column count -> scalar-load path
        flt = g[f'small{i}_filter'].tolist() if g[f'small{i}_has_filter'] else None
        got = ctx.segment_votes(g['small_votes'], 4, float(g[f'small{i}_threshold']), flt)
        assert np.array_equal(got, g[f'small{i}_classes']), i


def test_vote_uv2pt_errors_leave_votes_untouched(ctx):
    votes = np.zeros((4, 3))
    with pytest.raises(IndexError):
        ctx.vote_uv2pt(votes, np.array([0, 1], np.int32), np.array([0, 3], np.uint8))
    with pytest.raises(IndexError):
        ctx.vote_uv2pt(votes, np.array([1, 4], np.int32), np.array([0, 0], np.uint8))
    assert votes.sum() == 0
    ctx.vote_uv2pt(votes, np.array([-2, -1, 2, 2], np.int32), np.array([1, 1, 1, 2], np.uint8))
    assert votes[2, 1] == 1 and votes[2, 2] == 1 and votes.sum() == 2


def test_vote_and_segment_larger_random(ctx):
    rng = np.random.default_rng(11)
    npts, hw, ncols = 20000, 192 * 256, 134
    votes = np.zeros((npts, ncols)); want = np.zeros((npts, ncols))
    for f in range(6):
        lut = rng.integers(-1, npts, hw).astype(np.int32)
        lut[rng.random(hw) < 0.4] = -1
        lut[1000:3000] = lut[1000]                                   # heavy duplication
        mask = rng.choice(synth.ALPHABET, hw)
        ctx.vote_uv2pt(votes, lut, mask)
        O.vote_frame(want, lut, mask)
    assert np.array_equal(votes, want)
    for thr, flt in [(0.5, None), (0.5, [86, 114, 115]), (0.1, list(range(100, 134)))]:
        assert np.array_equal(ctx.segment_votes(votes, 133, thr, flt), O.segment(want, 133, thr, flt))


def _patch_luts(rng, nframes, h, w, npts, patch=5):
    """Lookups as Fusion.fuse writes them: every fused point owns a patch of neighbouring pixels (heavy duplication per frame)."""
    luts = np.full((nframes, h, w), -1, np.int32)
    for f in range(nframes):
        cells = rng.integers(0, npts, ((h + patch - 1) // patch, (w + patch - 1) // patch)).astype(np.int32)
        lut = np.repeat(np.repeat(cells, patch, 0), patch, 1)[:h, :w].copy()
        lut[rng.random((h, w)) < 0.3] = -1
        luts[f] = lut
    return luts.reshape(nframes, -1)


def test_vote_uv2pt_batch_equals_frame_by_frame(ctx, golden):
    """f3d_vote_uv2pt_batch*: the whole loop of VotingSegmentation.vote in one call -- equal to the reference's votes (voting.npz),
    to the oracle frame by frame on random and patch-structured lookups, across launch chunks (> 1023 frames), on repeated calls
    (generation-stamped set, never cleared) and interleaved with the per-frame entry point."""
    g = golden('voting')
    votes = np.zeros_like(g['votes'])
    h, w = g['masks'].shape[1:]
    ctx.vote_uv2pt_batch(votes, g['uv2pt'], g['masks'].reshape(len(g['masks']), -1), h, w)
    assert np.array_equal(votes, g['votes'])
    rng = np.random.default_rng(12)
    npts, h, w, ncols, F = 20000, 96, 130, 134, 7                      # w not a multiple of the 32-pixel tile
    for kind in ('random', 'patch'):
        luts = rng.integers(-1, npts, (F, h * w)).astype(np.int32) if kind == 'random' else _patch_luts(rng, F, h, w, npts)
        luts[2, 100:400] = luts[2, 100]                                # one pair many times
        luts[3, :5] = [-2, -1, -npts, npts - 1, 0]                     # NumPy negative indices wrap
        masks = rng.choice(synth.ALPHABET, (F, h * w)).astype(np.uint8)
        want = np.zeros((npts, ncols))
        for f in range(F):
            O.vote_frame(want, luts[f], masks[f])
        got = np.zeros((npts, ncols))
        ctx.vote_uv2pt_batch(got, luts, masks, h, w)
        assert np.array_equal(got, want), kind
        ctx.vote_uv2pt_batch(got, luts, masks, h, w)                   # again on the same context: adds the same increments once more
        assert np.array_equal(got, 2 * want), kind
        ctx.vote_uv2pt(got, luts[0], masks[0])                         # the per-frame call in between (it clears the shared set)
        ctx.vote_uv2pt_batch(got, luts[1:], masks[1:], h, w)
        assert np.array_equal(got, 3 * want), kind
    # more frames than one launch takes (1023): small frames
    F, h, w, npts = 1100, 40, 48, 3000
    luts = _patch_luts(rng, F, h, w, npts, patch=3)
    masks = rng.integers(0, 6, (F, h * w)).astype(np.uint8)
    want = np.zeros((npts, 6))
    for f in range(F):
        O.vote_frame(want, luts[f], masks[f])
    got = np.zeros((npts, 6))
    ctx.vote_uv2pt_batch(got, luts, masks, h, w)
    assert np.array_equal(got, want) and want.max() > 50


def test_vote_uv2pt_batch_index_error_keeps_the_earlier_frames(ctx):
    rng = np.random.default_rng(13)
    npts, h, w, F = 500, 33, 40, 6
    luts = rng.integers(-1, npts, (F, h * w)).astype(np.int32)
    masks = rng.integers(0, 4, (F, h * w)).astype(np.uint8)
    for bad_kind in ('label', 'point'):
        l2, m2 = luts.copy(), masks.copy()
        if bad_kind == 'label':
            m2[3, 77] = 9; l2[3, 77] = 5                               # label 9 >= 4 columns, on a valid lookup
        else:
            l2[3, 77] = npts
        want = np.zeros((npts, 4))
        for f in range(3):                                             # NumPy raises at frame 3: frames 0-2 are applied, 3-5 never run
            O.vote_frame(want, l2[f], m2[f])
        got = np.zeros((npts, 4))
        with pytest.raises(IndexError):
            ctx.vote_uv2pt_batch(got, l2, m2, h, w)
        assert np.array_equal(got, want), bad_kind
        ctx.vote_uv2pt_batch(got, luts[:3], masks[:3], h, w)           # the context is usable again
        assert np.array_equal(got, 2 * want)
    # many launches: the error flag used to be raised by blocks of the vote kernel itself, and blocks that started after it returned
    # without casting the votes of the frames BEFORE the offending one -- an intermittent wrong count (about one run in four)
    for it in range(60):
        F = 4 + it % 9
        lu = rng.integers(-1, npts, (F, h * w)).astype(np.int32)
        mk = rng.integers(0, 4, (F, h * w)).astype(np.uint8)
        fbad = int(rng.integers(0, F))
        mk[fbad, 5] = 200; lu[fbad, 5] = 7
        want = np.zeros((npts, 4))
        for f in range(fbad):
            O.vote_frame(want, lu[f], mk[f])
        got = np.zeros((npts, 4))
        with pytest.raises(IndexError):
            ctx.vote_uv2pt_batch(got, lu, mk, h, w)
        assert np.array_equal(got, want), (it, fbad)


def _dev_fuse(ctx, pts, views, masks, flt, thr, flags, presort=False, f32=False, nclasses=133, mask_shift=0, votes_at=None):
    """One call of f3d_project_vote_argmax_dev on device-resident inputs.  votes_at: caller-order indices whose uint16 vote
    rows are returned as well (the full [n, nclasses + 1] matrix stays on the device)."""
    import torch
    dev = torch.device('cuda', 0)
    x = torch.from_numpy(pts.astype(np.float32) if f32 else pts).to(dev)
    vd = torch.from_numpy(views).to(dev)
    mbuf = torch.zeros(masks.size + mask_shift, dtype=torch.uint8, device=dev)      # mask_shift: misaligned mask pointer
    md = mbuf[mask_shift:].view(masks.shape)
    md.copy_(torch.from_numpy(masks))
    n = len(pts)
    cls = torch.full((n,), -7, dtype=torch.int64, device=dev)
    votes = torch.full((n, nclasses + 1), 0xFFFF, dtype=torch.uint16, device=dev) if votes_at is not None else None
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))        # the fills / copies above run on the current stream: torch streams do not wait for it by themselves
    dt = f3d.F32 if f32 else f3d.F64
    with torch.cuda.stream(s):
        perm_ptr = None
        if presort:
            xs = torch.empty_like(x)
            perm = torch.empty(n, dtype=torch.int32, device=dev)
            ctx.cloud_sort_cells_dev(x.data_ptr(), dt, n, xs.data_ptr(), perm.data_ptr(), s.cuda_stream)
            s.synchronize()
            if n <= 20_000_000:
                assert np.array_equal(np.sort(perm.cpu().numpy()), np.arange(n))          # a permutation
                assert np.array_equal(xs.cpu().numpy(), x.cpu().numpy()[perm.cpu().numpy()])
            x, perm_ptr = xs, perm.data_ptr()
        V, H, W = masks.shape
        ctx.project_vote_argmax_dev(x.data_ptr(), dt, n, vd.data_ptr(), V, md.data_ptr(), H, W, nclasses, thr, flt,
                                    cls.data_ptr(), None if votes is None else votes.data_ptr(), s.cuda_stream, flags=flags, perm_ptr=perm_ptr)
        ctx.take_device_error(s.cuda_stream)
        s.synchronize()
    if votes_at is None:
        return cls.cpu().numpy()
    rows = votes.view(torch.int16)[torch.from_numpy(np.asarray(votes_at, np.int64)).to(dev)].cpu().numpy().view(np.uint16)
    return cls.cpu().numpy(), rows


def test_device_api_sort_flags_and_prepared_layout(ctx):
    sc = synth.scene('C1', n=100_000, mask_kind='iid')
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                 133, 0.5, None)
    for flags in (0, f3d.FUSE_SORT):
        assert np.array_equal(_dev_fuse(ctx, sc['points'], views, sc['masks'], None, 0.5, flags), want), flags
    assert np.array_equal(_dev_fuse(ctx, sc['points'], views, sc['masks'], None, 0.5, 0, presort=True), want)
    assert np.array_equal(_dev_fuse(ctx, sc['points'], views, sc['masks'], None, 0.5, f3d.FUSE_SORT, f32=True), want)
    # host entry point sorts clouds of this size by itself
    assert np.array_equal(ctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, 0.5, None), want)
    want_f = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                   133, 0.5, [86, 114, 115])
    got, votes = ctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, 0.5, [86, 114, 115], return_votes=True)
    assert np.array_equal(got, want_f)
    assert np.array_equal(votes.astype(np.float64),
                          O.forward_votes(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth']))


@pytest.mark.parametrize('mask_kind', ['block64', 'iid'])
def test_partially_deferred_points_parked_bins_and_their_overflow(ctx, mask_kind, monkeypatch):
    """A point the float32 kernel cannot finish keeps the votes of its proven views: bins parked in HBM, one mask of open views, the
    float64 tier visits those views only (calls with at most 64 views).  Deferred points beyond the parked slots (n/16 + 4096 of them) are
    redone from nothing -- forced here through F3D_DEBUG_PARK_SLOTS; an unsorted cloud (whole-scene wave boxes, wide float32 bounds)
    makes the deferred list long.  Labels and vote rows must not depend on which way a point went."""
    sc = synth.scene('C3', n=120_000, mask_kind=mask_kind)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    sel = np.arange(0, 120_000, 7)
    want, want_votes = O.project_vote_argmax(sc['points'][sel], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'],
                                             133, 0.0, None, return_votes=True)
    runs = {}
    for slots, flags in ((None, 0), (None, f3d.FUSE_SORT), ('100', 0), ('0', 0), ('100', f3d.FUSE_SORT)):
        if slots is None:
            monkeypatch.delenv('F3D_DEBUG_PARK_SLOTS', raising=False)
        else:
            monkeypatch.setenv('F3D_DEBUG_PARK_SLOTS', slots)
        cls, rows = _dev_fuse(ctx, sc['points'], views, sc['masks'], None, 0.0, flags, votes_at=sel)
        deferred = ctx.fuse_deferred()
        runs[(slots, flags)] = (cls, deferred)
        assert np.array_equal(cls[sel], want), (slots, flags)
        assert np.array_equal(rows.astype(np.float64), want_votes), (slots, flags)
    monkeypatch.delenv('F3D_DEBUG_PARK_SLOTS', raising=False)
    ref = runs[(None, 0)][0]
    for key, (cls, deferred) in runs.items():
        assert np.array_equal(cls, ref), key
    assert runs[('100', 0)][1][0] > 1000 and runs[(None, 0)][1][0] == runs[('100', 0)][1][0]     # the overflow path really ran
    assert (want != 133).mean() > 0.5


def test_general_K_matches(ctx):
    sc = synth.scene('C1', n=20000)
    K = np.array([[410.5, 1.75, 250.25], [0.125, 395.0, 260.5], [0., 0., 1.]])
    views = f3d.views_build(K, sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    want = O.project_vote_argmax(sc['points'], K, sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'])
    assert np.array_equal(ctx.project_vote_argmax(sc['points'], views, sc['masks']), want)


def test_sort_handles_nonfinite_and_degenerate_clouds(ctx):
    sc = synth.scene('C1', n=70000)
    pts = sc['points'].copy()
    pts[::1000] = np.nan
    pts[1::1000, 0] = np.inf
    pts[5] = 1e308
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    with np.errstate(all='ignore'):
        want = O.project_vote_argmax(pts, sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'])
    assert np.array_equal(ctx.project_vote_argmax(pts, views, sc['masks']), want)
    same = np.repeat(sc['points'][:1], 70000, axis=0)                          # zero-extent bounding box
    want = O.project_vote_argmax(same, sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'])
    assert np.array_equal(ctx.project_vote_argmax(same, views, sc['masks']), want)


def _near_pixel_boundary_points(K, q, t, rng, count, w, h):
    """World points whose exact projection lies within a few ulp of an integer pixel coordinate."""
    R = np.stack([O.rotate(q, np.eye(3)[k:k + 1])[0] for k in range(3)], axis=1)      # camera->world of unit axes
    u = rng.integers(1, w - 1, count).astype(np.float64)
    v = rng.uniform(1, h - 1, count)
    z = rng.uniform(0.3, 9.0, count)
    cam = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], axis=1)
    return cam @ R.T / np.dot(q, q) + t


def test_fast_projection_and_f32_cull_never_disagree_with_exact(ctx):
    rng = np.random.default_rng(77)
    sc = synth.scene('C3', n=1)
    q, t, K = sc['wxyzs'], sc['translations'], sc['K']
    q = q.copy(); q[3] *= 1.7; q[5] *= -0.4                                    # un-normalised poses too
    views = f3d.views_build(K, sc['w'], sc['h'], q, t, sc['max_depth'])
    clouds = [synth.cloud(400_000),                                            # f32-representable
              rng.uniform([-5, -5, 0], [5, 5, 3], (200_000, 3)),               # full f64 mantissas
              t[rng.integers(0, len(t), 50_000)] + rng.normal(size=(50_000, 3)) * 1e-3,   # hugging the eyes (z -> 0)
              np.vstack([_near_pixel_boundary_points(K, q[j], t[j], rng, 4000, sc['w'], sc['h']) for j in range(0, 64, 4)])]
    fallbacks = decided = 0
    for k, pts in enumerate(clouds):
        for dim in (1024, 16):                                                 # a tiny image exercises the out-of-image rule as well
            pairs, fb, wrong, cullwrong = ctx.fastpath_audit(pts, views, dim, dim)
            assert pairs > 0 and wrong == 0 and cullwrong == 0, (k, dim, pairs, fb, wrong, cullwrong)
            fallbacks += fb; decided += pairs - fb
    assert fallbacks > 0 and decided > 0                                       # both branches really run
    pairs, fb, wrong, cullwrong = ctx.fastpath_audit(synth.cloud(4_000_000), views, 1024, 1024)   # a cloud as dense as C3's cells assume
    assert wrong == 0 and cullwrong == 0 and fb < 0.01 * pairs, (pairs, fb)    # per pair the float32 bound leaves < 1 % undecided
    # ... and the fused kernel agrees with the oracle on the adversarial sets
    pts = np.vstack([c[:20000] for c in clouds])
    with np.errstate(all='ignore'):
        want = O.project_vote_argmax(pts, K, q, t, sc['masks'], sc['max_depth'])
    assert np.array_equal(ctx.project_vote_argmax(pts, views, sc['masks']), want)


def test_views_record_fast_operator_matches_canonical_rotation():
    sc = synth.scene('C2', n=1)
    q = sc['wxyzs'].copy(); q[2] *= 3.0
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], q, sc['translations'], 10.0)
    F = f3d.view_fields(views)
    d = np.random.default_rng(1).normal(size=(100, 3))
    for j in range(len(q)):
        c = O.rotate(F['qinv'][j], d)
        want = (sc['K'] @ c.T).T
        got = d @ F['M'][j].T
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
        assert (F['mnorm'][j] >= np.abs(sc['K']).sum(1) * np.dot(F['qinv'][j], F['qinv'][j])).all()


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json configurations at full size: oracle on a subset + size-independent properties
# ------------------------------------------------------------------------------------------------------------
SETTINGS = [(0.5, None), (0.0, None), (0.5, [86, 114, 115])]          # (threshold, filter_classes)


def _full_size_case(ctx, name, mask_kind, subset, n=None, light=False):
    """A BASELINE.json configuration at its full size.  The oracle labels a random subset of the points (labels and votes
    are per-point functions, so the full run must agree at those indices), at ALL THREE settings of SURVEY 8(d) and with
    the complete vote rows compared; then size-independent properties of the full result.  Returns {setting: labels}."""
    sc = synth.scene(name, mask_kind=mask_kind, n=n)
    pts, n = sc['points'], len(sc['points'])
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    rng = np.random.default_rng(123)
    idx = np.sort(rng.choice(n, subset, replace=False))
    want_votes = O.forward_votes(pts[idx], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], ncols=134)
    assert (want_votes.sum(1) > 0).mean() > 0.5                                 # most points are seen by some view
    out = {}
    for k, (thr, flt) in enumerate(SETTINGS):
        want = O.segment(want_votes, 133, thr, flt)
        if k == 1:                                                              # the vote rows themselves, once per configuration
            labels, rows = _dev_fuse(ctx, pts, views, sc['masks'], flt, thr, f3d.FUSE_SORT, votes_at=idx)
            assert np.array_equal(rows.astype(np.float64), want_votes)
        else:
            labels = _dev_fuse(ctx, pts, views, sc['masks'], flt, thr, f3d.FUSE_SORT)
        assert labels.min() >= 0 and labels.max() <= 133
        assert np.array_equal(labels[idx], want), (name, thr, flt, int((labels[idx] != want).sum()))
        out[(thr, None if flt is None else tuple(flt))] = labels
    # the comparison is not "unclassified == unclassified": at threshold 0 every point that was sampled carries a real label
    real = (O.segment(want_votes, 133, 0.0, None) != 133).mean()
    assert real > 0.5, real
    thr, flt = SETTINGS[1]
    labels = out[(thr, None)]
    assert abs((labels != 133).mean() - real) < 0.02
    # (2) sorted-in-call, caller-order and prepared-layout paths agree everywhere
    if not light:
        assert np.array_equal(_dev_fuse(ctx, pts, views, sc['masks'], flt, thr, 0), labels)
    assert np.array_equal(_dev_fuse(ctx, pts, views, sc['masks'], flt, thr, 0, presort=True), labels)
    # (3) permutation equivariance: labelling a shuffled cloud = shuffling the labels
    perm = rng.permutation(n)
    assert np.array_equal(_dev_fuse(ctx, pts[perm], views, sc['masks'], flt, thr, f3d.FUSE_SORT), labels[perm])
    del perm
    # (4) float32 storage of the (f32-representable) cloud gives the same labels
    assert np.array_equal(_dev_fuse(ctx, pts, views, sc['masks'], flt, thr, f3d.FUSE_SORT, f32=True), labels)
    # (5) a view that sees nothing new changes nothing: appending a camera that looks away from the cloud
    if not light:
        q_far, t_far = synth.ring_views(1)
        t_far = t_far + np.array([1000.0, 0, 0])
        views2 = f3d.views_build(sc['K'], sc['w'], sc['h'], np.vstack([sc['wxyzs'], q_far]), np.vstack([sc['translations'], t_far]), sc['max_depth'])
        masks2 = np.concatenate([sc['masks'], np.full((1,) + sc['masks'].shape[1:], 7, np.uint8)])
        assert np.array_equal(_dev_fuse(ctx, pts, views2, masks2, flt, thr, f3d.FUSE_SORT), labels)
    return out, sc, views


def test_config_c2_1m_points_16_rtab_views(ctx):
    out, _, _ = _full_size_case(ctx, 'C2', 'block64', 60_000)
    assert (out[(0.5, (86, 114, 115))] != 133).mean() > 0.05


def test_config_c2_iid_masks(ctx):
    _full_size_case(ctx, 'C2', 'iid', 30_000, light=True)


def test_config_c3_10m_points_64_views(ctx):
    out, _, _ = _full_size_case(ctx, 'C3', 'block64', 60_000)
    assert (out[(0.0, None)] != 133).mean() > 0.8            # threshold 0: the label of every sampled point is a real plurality
    assert (out[(0.5, None)] != 133).sum() > 1000            # independent random masks rarely give a 50 % majority; a few points do


def test_config_c5_50m_points_256_views(ctx):
    """C5's fused leg: 50M points x 256 views x 1024^2 masks (256 MiB of masks, 13.4 GB of uint16 votes on the device).
    V > 255: the fast kernel's 8-bit bins can overflow, which sends a point to the exact kernel's 16-bit bins -- forced
    below with masks that agree in every view, on a slice of the cloud."""
    out, sc, views = _full_size_case(ctx, 'C5', 'block64', 20_000, light=True)
    assert (out[(0.0, None)] != 133).mean() > 0.8
    pts = sc['points'][:2_000_000]
    masks = np.full(sc['masks'].shape, 86, np.uint8)
    masks[:, :256, :] = sc['masks'][:, :256, :]                     # the top quarter of every image keeps its blocks
    idx = np.arange(0, len(pts), 100)
    want_votes = O.forward_votes(pts[idx], sc['K'], sc['wxyzs'], sc['translations'], masks, sc['max_depth'], ncols=134)
    assert want_votes.max() > 255                                   # some point is seen as 86 by every one of the 256 views
    labels, rows = _dev_fuse(ctx, pts, views, masks, None, 0.5, f3d.FUSE_SORT, votes_at=idx)
    assert np.array_equal(rows.astype(np.float64), want_votes)
    assert np.array_equal(labels[idx], O.segment(want_votes, 133, 0.5, None))


def test_config_c1_full_vs_oracle(ctx):
    sc = synth.scene('C1', mask_kind='iid')
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    for thr, flt in [(0.5, None), (0.0, None), (0.5, [86, 114, 115])]:
        want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, thr, flt)
        assert np.array_equal(ctx.project_vote_argmax(sc['points'], views, sc['masks'], 133, thr, flt), want)


def test_mask_coding_any_size_and_alignment(ctx):
    """k_code_masks: 8x8 tiles of bin codes for any H, W (padded tiles) and any mask pointer alignment."""
    rng = np.random.default_rng(8)
    q, t = synth.ring_views(6)
    pts = synth.cloud(40_000)
    for (h, w, shift) in [(60, 100, 0), (61, 99, 0), (64, 104, 3), (64, 104, 0), (1, 8, 0), (3, 5, 1)]:
        K = np.array([[w * 0.8, 0, w / 2], [0, w * 0.8, h / 2], [0, 0, 1]])
        masks = rng.integers(0, 134, (6, h, w), dtype=np.uint8)
        views = f3d.views_build(K, w, h, q, t, 10.0)
        want = O.project_vote_argmax(pts, K, q, t, masks, 10.0, 133, 0.3, None)
        for flags in (0, f3d.FUSE_SORT):
            assert np.array_equal(_dev_fuse(ctx, pts, views, masks, None, 0.3, flags, mask_shift=shift), want), (h, w, flags)


def test_nclasses_edge_values(ctx):
    """nclasses = 253 is the last value with byte codes to spare (fast kernel); 254 and 255 run on the exact kernel alone;
    a label above nclasses raises IndexError only when such a pixel is sampled."""
    rng = np.random.default_rng(9)
    sc = synth.scene('C1', n=30_000)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    masks = rng.integers(0, 256, sc['masks'].shape, dtype=np.uint8)
    for ncls in (255, 254):
        m = np.minimum(masks, ncls)
        want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], m, sc['max_depth'], ncls, 0.3, None)
        for flags in (0, f3d.FUSE_SORT):
            assert np.array_equal(_dev_fuse(ctx, sc['points'], views, m, None, 0.3, flags, nclasses=ncls), want), (ncls, flags)
    with pytest.raises(IndexError):                                    # exact-kernel-only path reports it as well
        _dev_fuse(ctx, sc['points'], views, masks, None, 0.3, 0, nclasses=254)
    masks253 = np.minimum(masks, 253)
    want = O.project_vote_argmax(sc['points'], sc['K'], sc['wxyzs'], sc['translations'], masks253, sc['max_depth'], 253, 0.3, None)
    assert np.array_equal(_dev_fuse(ctx, sc['points'], views, masks253, None, 0.3, f3d.FUSE_SORT, nclasses=253), want)
    with pytest.raises(IndexError):
        _dev_fuse(ctx, sc['points'], views, masks, None, 0.3, f3d.FUSE_SORT, nclasses=253)
    # labels above nclasses that no point samples are harmless (voting.py:98 only sees sampled pixels)
    pts = sc['points'][:2000]
    loud = np.minimum(masks, 100)
    loud[:, 0, 0] = 200
    votes = O.forward_votes(pts, sc["K"], sc["wxyzs"], sc["translations"], loud, sc["max_depth"], ncols=256)
    assert votes[:, 200].sum() == 0                                    # nobody looks at pixel (0, 0)
    want = O.segment(votes[:, :101], 100, 0.3, None)
    assert np.array_equal(_dev_fuse(ctx, pts, views, loud, None, 0.3, 0, nclasses=100), want)


# ---- the view-chunked call (f3d_fuse_chunked_begin_dev / f3d_fuse_chunk_dev): same labels as the one-shot call ----------------
def _dev_fuse_chunked(ctx, pts, views, masks, flt, thr, flags, bounds, f32=False, nclasses=133, order=None, presence='own', coded=False):
    """bounds: chunk boundaries [0, ..., V].  order: views (and masks) handed over in this order (a permutation of range(V)).
    coded: the planes are coded up front with f3d_code_planes_dev (as another rank would) and handed over coded."""
    import torch
    dev = torch.device('cuda', 0)
    if order is not None:
        views, masks = views[order], masks[order]
    x = torch.from_numpy(pts.astype(np.float32) if f32 else pts).to(dev)
    vd, md = torch.from_numpy(np.ascontiguousarray(views)).to(dev), torch.from_numpy(np.ascontiguousarray(masks)).to(dev)
    n, (V, H, W) = len(pts), masks.shape
    cls = torch.full((n,), -7, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream(dev)
    dt = f3d.F32 if f32 else f3d.F64
    present = torch.empty(256, dtype=torch.uint8, device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))        # (see _dev_fuse)
    with torch.cuda.stream(s):
        if presence == 'own':
            ctx.mask_presence_dev(md.data_ptr(), V, H, W, present.data_ptr(), s.cuda_stream)
        ctx.fuse_chunked_begin_dev(present.data_ptr() if presence == 'own' else None, n, V, H, W, nclasses, flt, s.cuda_stream)
        if coded:
            cd = torch.full((V, ctx.coded_plane_bytes(H, W)), 0xEE, dtype=torch.uint8, device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            for a, b in zip(bounds[:-1], bounds[1:]):                                   # coded chunk by chunk, like the ranks do
                ctx.code_planes_dev(md[a:b].data_ptr(), b - a, H, W, cd[a:b].data_ptr(), s.cuda_stream)
            md.fill_(0xEE)                                                              # the raw masks are gone: nothing may read them
            s.wait_stream(torch.cuda.current_stream(dev))
        for a, b in zip(bounds[:-1], bounds[1:]):
            if coded:
                ctx.fuse_chunk_coded_dev(x.data_ptr(), dt, n, vd.data_ptr(), V, a, b, cd.data_ptr(), H, W, nclasses, thr, flt, cls.data_ptr(),
                                         s.cuda_stream, flags=flags)
            else:
                ctx.fuse_chunk_dev(x.data_ptr(), dt, n, vd.data_ptr(), V, a, b, md.data_ptr(), H, W, nclasses, thr, flt, cls.data_ptr(),
                                   s.cuda_stream, flags=flags)
        ctx.take_device_error(s.cuda_stream)
        s.synchronize()
    if presence == 'own' and not coded:
        got = np.flatnonzero(present.cpu().numpy())
        assert np.array_equal(got, np.unique(masks)), 'labels present'
    return cls.cpu().numpy()


@pytest.mark.parametrize('mask_kind', ['block64', 'block64x40', 'iid'])
def test_view_chunked_call_equals_the_one_shot_call(ctx, mask_kind):
    """Every instance of the fast kernel (dword bins, packed bins, any-alphabet) resumed from the carry in HBM: chunk splits of
    all shapes, the views in a shuffled order, float32 / float64 clouds, both thresholds, a filter list -- labels identical to
    f3d_project_vote_argmax_dev and to the oracle."""
    V = 24
    sc = synth.scene('C1', n=150_001, mask_kind=mask_kind)
    q, t = synth.ring_views(V)
    masks = synth.masks(V, sc['h'], sc['w'], mask_kind)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], q, t, sc['max_depth'])
    pts = sc['points']
    rng = np.random.default_rng(3)
    sub = rng.choice(len(pts), 3000, replace=False)
    for thr, flt, f32 in [(0.0, None, False), (0.5, None, True), (0.5, [86, 114, 115], False)]:
        one = _dev_fuse(ctx, pts, views, masks, flt, thr, f3d.FUSE_SORT, f32=f32)
        p_sub = pts[sub].astype(np.float32).astype(np.float64) if f32 else pts[sub]
        want = O.project_vote_argmax(p_sub, sc['K'], q, t, masks, sc['max_depth'], 133, thr, flt)
        assert np.array_equal(one[sub], want)
        if thr == 0.0:
            assert (one != 133).mean() > 0.3                       # real labels, not the 'unclassified' default
        for bounds, order, flags, presence in [([0, V], None, f3d.FUSE_SORT, 'own'),
                                               ([0, 8, 16, V], None, f3d.FUSE_SORT, 'own'),
                                               ([0, 1, 2, 3, 17, V], rng.permutation(V), f3d.FUSE_SORT, 'own'),
                                               ([0, 12, V], rng.permutation(V), 0, 'all'),
                                               (list(range(V + 1)), None, f3d.FUSE_SORT, 'all')]:
            got = _dev_fuse_chunked(ctx, pts, views, masks, flt, thr, flags, bounds, f32=f32, order=order, presence=presence)
            assert np.array_equal(got, one), (mask_kind, thr, flt, f32, bounds)
        # the coded exchange (f3d_code_planes_dev + f3d_fuse_chunk_coded_dev): the raw masks are destroyed before the first chunk votes
        for bounds, order, presence in [([0, 8, 16, V], None, 'own'), ([0, V], rng.permutation(V), 'all'), ([0, 5, V], None, 'own')]:
            got = _dev_fuse_chunked(ctx, pts, views, masks, flt, thr, f3d.FUSE_SORT, bounds, f32=f32, order=order, presence=presence, coded=True)
            assert np.array_equal(got, one), ('coded', mask_kind, thr, flt, f32, bounds)


def test_view_chunked_call_deferred_points_errors_and_sequence(ctx):
    import torch
    dev = torch.device('cuda', 0)
    sc = synth.scene('C1', n=60_000)
    views = f3d.views_build(sc['K'], sc['w'], sc['h'], sc['wxyzs'], sc['translations'], sc['max_depth'])
    pts = sc['points'].copy()
    pts[::97] *= 1e31                       # not representable in the float32 kernel: deferred in every chunk, labelled by the exact tier
    pts[5::101] = np.nan
    V = len(views)
    one = _dev_fuse(ctx, pts, views, sc['masks'], None, 0.0, f3d.FUSE_SORT)
    got = _dev_fuse_chunked(ctx, pts, views, sc['masks'], None, 0.0, f3d.FUSE_SORT, [0, 1, 3, V])
    assert np.array_equal(got, one)
    assert ctx.fuse_deferred()[0] >= len(pts[::97])
    # the same through the coded exchange: the deferred points end in the reference-arithmetic tier that reads CODED planes
    got = _dev_fuse_chunked(ctx, pts, views, sc['masks'], None, 0.0, f3d.FUSE_SORT, [0, 1, 3, V], coded=True)
    assert np.array_equal(got, one) and ctx.fuse_deferred()[1] >= len(pts[::97])
    want = O.project_vote_argmax(sc['points'][:3000], sc['K'], sc['wxyzs'], sc['translations'], sc['masks'], sc['max_depth'], 133, 0.0, None)
    huge = sc['points'][:3000] * 1.0
    assert np.array_equal(_dev_fuse_chunked(ctx, huge, views, sc['masks'], None, 0.0, 0, [0, 2, V], coded=True), want)
    # a rejected label in a middle chunk: IndexError once the call is complete, like the one-shot call
    bad = sc['masks'].copy(); bad[1, 100:300, 100:300] = 200
    with pytest.raises(IndexError, match='project_vote_argmax'):
        _dev_fuse_chunked(ctx, sc['points'], views, bad, None, 0.0, 0, [0, 1, 2, V])
    with pytest.raises(IndexError, match='project_vote_argmax'):
        _dev_fuse_chunked(ctx, sc['points'], views, bad, None, 0.0, 0, [0, 1, 2, V], coded=True)
    # chunks out of sequence, a chunk without begin, more than 255 views
    x = torch.from_numpy(sc['points']).to(dev); vd = torch.from_numpy(views).to(dev); md = torch.from_numpy(sc['masks']).to(dev)
    cls = torch.empty(len(pts), dtype=torch.int64, device=dev)
    H, W = sc['masks'].shape[1:]
    args = (x.data_ptr(), f3d.F64, len(pts), vd.data_ptr(), V)
    tail = (md.data_ptr(), H, W, 133, 0.0, None, cls.data_ptr(), None)
    ctx.fuse_chunked_begin_dev(None, len(pts), V, H, W, 133, None, None)
    with pytest.raises(ValueError):
        ctx.fuse_chunk_dev(*args, 1, 2, *tail)
    ctx.fuse_chunk_dev(*args, 0, 2, *tail)
    with pytest.raises(ValueError):
        ctx.fuse_chunk_dev(*args, 0, 2, *tail)
    ctx.fuse_chunk_dev(*args, 2, V, *tail)
    with pytest.raises(ValueError):
        ctx.fuse_chunk_dev(*args, 0, V, *tail)            # the call is complete: begin again first
    with pytest.raises(ValueError):
        ctx.fuse_chunked_begin_dev(None, len(pts), 256, H, W, 133, None, None)
    ctx.synchronize()
    assert np.array_equal(cls.cpu().numpy(), _dev_fuse(ctx, sc['points'], views, sc['masks'], None, 0.0, 0))


def test_fused_randomised_configurations(ctx):
    """scripts/fused_fuzz.py: random scene / view count (incl. > 64 and > 255) / mask size / label alphabet (every k_fuse instance) /
    threshold / filter / dtype / sort flag / chunk split -- labels and vote rows against the oracle, chunked against one-shot."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location('fused_fuzz', Path(__file__).resolve().parent.parent / 'scripts' / 'fused_fuzz.py')
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = np.random.default_rng(77)
    for k in range(40):
        fuzz.one_config(ctx, rng, k, verbose=False)


def test_rotate_dev_equals_host_pointer_rotate(ctx):
    import torch
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(100_003, 3)) * 7.0
    q = np.array([0.3, -0.4, 0.5, 0.7])                          # not normalised, like the reference's inverse quaternions
    want = O.rotate(q, pts)
    x = torch.from_numpy(pts).to(dev); out = torch.empty_like(x)
    ctx.rotate_dev(x.data_ptr(), len(pts), q, out.data_ptr(), None)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(ctx.rotate(pts, q), want)


def test_vote_segment_merge_randomised_configurations(ctx):
    """scripts/aux_fuzz.py: batched vote (offending frames, negative and duplicate lookups), segment_votes (thresholds, filter lists) and
    merge_bb (random blob scenes; the oracle's fit injected with and without the candidate prefilter, or the product's own GPU fit) against the
    oracle; f3d_obb_fit on random, lattice and nearly flat point sets."""
    import contextlib
    import importlib.util
    import io
    from pathlib import Path
    spec = importlib.util.spec_from_file_location('aux_fuzz', Path(__file__).resolve().parent.parent / 'scripts' / 'aux_fuzz.py')
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    rng = np.random.default_rng(31)
    for k in range(60):
        fuzz.vote_config(ctx, rng)
        if k % 3 == 0:
            fuzz.obb_config(ctx, rng)                           # f3d_obb_fit: hull vertices against scipy's Qhull, boxes against the oracle's recipe
        if k % 6 == 0:
            with contextlib.redirect_stdout(io.StringIO()):
                fuzz.merge_config(rng)
