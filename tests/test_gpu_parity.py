"""GPU parity: every HIP stage against the CPU oracle on identical seeded inputs, through the C ABI.
Integer / byte outputs must be bit-exact; the 128-D embedding within L2 1e-4 (BASELINE.json north_star)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dump(name, **arrays):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    np.savez_compressed(os.path.join(d, name + ".npz"), **arrays)


def _detector(oracle):
    from pyannote_video_amd import models
    return oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))


def test_pyramid_levels_bit_exact(ctx, oracle, small_video):
    det = _detector(oracle)
    f = small_video.frame(0)
    up = oracle.Detector.pyramid_level(det, f, 1, 0)
    n = det.levels(up.shape[0], up.shape[1])
    assert n >= 5
    for l in range(n):
        a = ctx.pyramid_level(f, 1, l)
        b = det.pyramid_level(f, 1, l)
        assert a.shape == b.shape, (l, a.shape, b.shape)
        bad = int((a != b).sum())
        if bad:
            _dump("pyr_mismatch_l%d" % l, gpu=a, cpu=b)
        assert bad == 0, "level %d: %d differing bytes" % (l, bad)


@pytest.mark.parametrize("mode", ["mul", "accumulate"])
def test_pyramid_bit_exact_under_both_coordinate_generations(oracle, small_video, monkeypatch, mode):
    """oracle/EXT_REGISTER.md E1: whether dlib's resize_image forms a source coordinate as i * scale or carries it from index to index is
    recalled, not read.  Both the oracle (PVO_RESIZE_COORDS) and the HIP path (PVF_RESIZE_COORDS: host-built row and column tables, the
    kernel is the same) can do either; the pyramid bytes must agree under each."""
    from pyannote_video_amd import models
    from pyannote_video_amd.runtime import Context
    monkeypatch.setenv("PVO_RESIZE_COORDS", mode)
    monkeypatch.setenv("PVF_RESIZE_COORDS", mode)
    det = _detector(oracle)
    c = Context(device=0, detector=models.DEFAULT_DETECTOR)          # (a context of its own: the tables belong to a context's plans)
    try:
        f = small_video.frame(5)
        up = oracle.Detector.pyramid_level(det, f, 1, 0)
        n = det.levels(up.shape[0], up.shape[1])
        for l in range(n):
            a, b = c.pyramid_level(f, 1, l), det.pyramid_level(f, 1, l)
            assert a.shape == b.shape and int((a != b).sum()) == 0, "mode %s, level %d: %d differing bytes" % (mode, l, int((a != b).sum()))
    finally:
        c.close()


@pytest.mark.parametrize("cell,pad,shape", [(8, 10, (200, 264)), (8, 10, (97, 131)), (4, 1, (23, 23)), (1, 3, (64, 64)), (4, 1, (40, 52))])
def test_fhog_bit_exact(ctx, oracle, small_video, cell, pad, shape):
    f = small_video.frame(1)[40:40 + shape[0], 100:100 + shape[1]]
    a = ctx.fhog(f, cell, pad, pad)
    b = oracle.fhog(f, cell, pad, pad)
    assert a.shape == b.shape
    bad = int((a.view(np.uint32) != b.view(np.uint32)).sum())
    if bad:
        _dump("fhog_mismatch_c%d" % cell, gpu=a, cpu=b)
    assert bad == 0, "fhog cell %d: %d of %d floats differ (max abs %g)" % (cell, bad, a.size, np.abs(a - b).max())


def test_batched_detector_level_features_bit_exact(ctx, oracle, small_video):
    """the all-levels-per-launch FHOG kernels (what detect_batch runs) against the oracle, every cell of several levels"""
    f = small_video.frame(3)
    for level in (0, 1, 4, 9):
        img = ctx.pyramid_level(f, 1, level)
        a = ctx.level_features(f, 1, level)
        b = oracle.fhog(img, 8, 10, 10)
        assert a.shape == b.shape, (level, a.shape, b.shape)
        bad = int((a.view(np.uint32) != b.view(np.uint32)).sum())
        if bad:
            _dump("level_feat_mismatch_%d" % level, gpu=a, cpu=b)
        assert bad == 0, "level %d: %d of %d feature floats differ" % (level, bad, a.size)


def test_detector_raw_and_boxes_bit_exact(ctx, oracle, small_video):
    det = _detector(oracle)
    for i in (0, 7):
        f = small_video.frame(i)
        raw_g = ctx.detect_raw(f, 1)
        raw_c = det.detect_raw(f, 1)
        assert len(raw_c) > 0
        if raw_g != raw_c:
            _dump("raw_mismatch_%d" % i, gpu=np.array([(r[0],) + r[1:5] + r[5] for r in raw_g], np.float64),
                  cpu=np.array([(r[0],) + r[1:5] + r[5] for r in raw_c], np.float64))
        assert len(raw_g) == len(raw_c)
        assert raw_g == raw_c
        boxes_g, scores_g = ctx.detect(f, 1)
        fin_c = det.detect(f, 1)
        assert boxes_g == [d[5] for d in fin_c]
        assert np.array_equal(scores_g, np.array([d[0] for d in fin_c], np.float32))
        assert len(boxes_g) == small_video.faces


def test_detect_batch_equals_single(ctx, small_video):
    frames = [small_video.frame(i) for i in range(4)]
    single = [ctx.detect(f, 1)[0] for f in frames]
    batch = [b for b, _ in ctx.detect_batch(frames, 1)]
    assert single == batch


def test_detect_many_pipelined_equals_single(ctx, small_video):
    frames = [small_video.frame(i) for i in range(7)]
    single = [ctx.detect(f, 1) for f in frames]
    many = ctx.detect_many(frames, 3, 1)             # batches of 3, 3, 1 with two in flight
    assert [b for b, _ in many] == [b for b, _ in single]
    for (_, sa), (_, sb) in zip(many, single):
        assert np.array_equal(sa, sb)


def test_chips_bit_exact(ctx, oracle, small_video):
    f = small_video.frame(2)
    cases = [((100.5, 60.25, 180.75, 140.0), 1.0, 0.0, 64, 64),        # mild scale
             ((50.0, 30.0, 450.0, 330.0), 1.0, 0.0, 64, 64),           # two pyramid levels
             ((200.0, 100.0, 330.0, 230.0), 0.9659258262890683, 0.25881904510252074, 150, 150),  # rotated
             ((-40.0, -30.0, 120.0, 130.0), 1.0, 0.0, 64, 64),         # partly outside
             ((300.0, 200.0, 320.0, 220.0), 0.8, 0.6, 150, 150),       # upsampling chip
             ((1000.0, 1000.0, 1100.0, 1100.0), 1.0, 0.0, 64, 64)]     # fully outside
    for rect, cs, sn, rows, cols in cases:
        a = ctx.extract_chip(f, rect, cs, sn, rows, cols)
        b = oracle.extract_chip(f, rect, cs, sn, rows, cols)
        bad = int((a != b).sum())
        if bad:
            _dump("chip_mismatch", gpu=a, cpu=b)
        assert bad == 0, (rect, bad)


def test_landmarks_bit_exact(ctx, oracle, small_video, model_paths):
    from pyannote_video_amd import models
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    frames, boxes = [], []
    for i in (0, 3, 8):
        f = small_video.frame(i)
        for b in ctx.detect(f, 1)[0]:
            frames.append(f); boxes.append(b)
    boxes.append((-20, -10, 60, 70)); frames.append(small_video.frame(0))   # box leaving the frame
    pts = ctx.landmarks(frames, boxes)
    for k, (f, b) in enumerate(zip(frames, boxes)):
        ref = sp(f, b)
        assert np.array_equal(pts[k], ref), (k, b, np.abs(pts[k] - ref).max())


def test_face_chips_and_embedding(ctx, oracle, small_video, model_paths):
    from pyannote_video_amd import models
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    frames, pts = [], []
    for i in (0, 5):
        f = small_video.frame(i)
        for b in ctx.detect(f, 1)[0]:
            frames.append(f); pts.append(sp(f, b))
    chips = ctx.face_chips(frames, pts)
    ref_chips = np.stack([emb.chip(f, p) for f, p in zip(frames, pts)])
    assert np.array_equal(chips, ref_chips), int((chips != ref_chips).sum())
    out = ctx.embed(frames, pts)
    ref = np.stack([emb.forward(c) for c in ref_chips])
    err = np.linalg.norm(out - ref, axis=1)
    scale = np.linalg.norm(ref, axis=1)
    _dump("embed_check", gpu=out, cpu=ref)
    assert err.max() <= 1e-4, (err.max(), scale.mean())
    out2 = ctx.embed_chips(ref_chips)
    assert np.array_equal(out, out2)


def test_tracker_bit_exact(ctx, oracle, small_video):
    from pyannote_video_amd import models
    tabs = models.dsst_tables()
    f0 = small_video.frame(0)
    boxes = ctx.detect(f0, 1)[0]
    assert boxes
    ref = [oracle.Tracker(tabs) for _ in boxes]
    trk = [ctx.tracker_create() for _ in boxes]
    dbox = [tuple(float(v) for v in b) for b in boxes]
    ctx.tracker_start_many(trk, [f0] * len(trk), dbox)
    for r, b in zip(ref, dbox):
        r.start_track(f0, b)
    _, A, B = ctx.tracker_state(trk[0])             # start_track keeps the plane spectra on chip: A = G * F and B = sum |F|^2 pin them
    Ar, Br = ref[0].debug_state()
    assert np.array_equal(A, Ar), np.abs(A - Ar).max()
    assert np.array_equal(B, Br)
    for i in range(1, 5):
        f = small_video.frame(i)
        psr, pos = ctx.tracker_update_many(trk, [f] * len(trk))
        for k, r in enumerate(ref):
            p = r.update(f)
            assert psr[k] == p, (i, k, psr[k], p)
            assert tuple(pos[k]) == r.get_position(), (i, k, pos[k], r.get_position())
            assert ctx.tracker_position(trk[k]) == r.get_position()
        assert psr.min() > 5.0
        F, A, B = ctx.tracker_state(trk[0])         # the full update writes the spectra of this frame's features out
        Ar, Br = ref[0].debug_state()
        assert np.array_equal(F, ref[0].debug_F())
        assert np.array_equal(A, Ar) and np.array_equal(B, Br)
    for t in trk:
        ctx.tracker_destroy(t)


def test_tracker_deferred_update_and_commit_bit_exact(ctx, oracle, small_video):
    """update without the model update + commit on the same frame == immediate update (outputs and filter state), and both
    equal the oracle's tracker"""
    from pyannote_video_amd import models, _lib
    tabs = models.dsst_tables()
    f0 = small_video.frame(0)
    boxes = ctx.detect(f0, 1)[0]
    dbox = [tuple(float(v) for v in b) for b in boxes]
    n = len(dbox)
    ref = [oracle.Tracker(tabs) for _ in dbox]
    for r, b in zip(ref, dbox):
        r.start_track(f0, b)
    trk = ctx.tracker_create_many(n)
    ctx.tracker_start_many(trk, [f0] * n, dbox)
    f1, f2, f3 = small_video.frame(1), small_video.frame(2), small_video.frame(3)
    psr, pos = ctx.tracker_update_many(trk, [f1] * n, defer=True)
    want = [(r.update(f1), r.get_position()) for r in ref]
    assert [(psr[k], tuple(pos[k])) for k in range(n)] == want
    _, A0, B0 = ctx.tracker_state(trk[0])
    with pytest.raises(_lib.PvfError):                       # an uncommitted tracker refuses the next update
        ctx.tracker_update_many(trk[:1], [f2])
    ctx.tracker_commit_many(trk, [f1] * n)
    _, A1, B1 = ctx.tracker_state(trk[0])
    Ar, Br = ref[0].debug_state()
    assert not np.array_equal(A0, A1)                        # the deferred call had left the filters alone
    assert np.array_equal(A1, Ar) and np.array_equal(B1, Br)
    assert ctx.tracker_position(trk[0]) == ref[0].get_position()
    for f in (f2, f3):
        psr, pos = ctx.tracker_update_many(trk, [f] * n)
        for k, r in enumerate(ref):
            assert psr[k] == r.update(f)
            assert tuple(pos[k]) == r.get_position()
    ctx.tracker_destroy_many(trk)


def test_tracker_clone_bit_exact(ctx, oracle, small_video):
    """a cloned tracker behaves exactly like the one it was copied from (and like the oracle's)"""
    from pyannote_video_amd import models
    tabs = models.dsst_tables()
    f0, f1 = small_video.frame(4), small_video.frame(3)
    boxes = ctx.detect(f0, 1)[0]
    dbox = [tuple(float(v) for v in b) for b in boxes]
    n = len(dbox)
    trk = ctx.tracker_create_many(n)
    ctx.tracker_start_many(trk, [f0] * n, dbox)
    twin = ctx.tracker_clone_many(trk)
    _, A0, B0 = ctx.tracker_state(trk[0])
    _, A1, B1 = ctx.tracker_state(twin[0])
    assert np.array_equal(A0, A1) and np.array_equal(B0, B1)
    assert [ctx.tracker_position(t) for t in twin] == [ctx.tracker_position(t) for t in trk]
    # clones share their source's filters until one side writes them: deferred updates of both IN ONE CALL only read them ...
    pd, bd = ctx.tracker_update_many(trk + twin, [f1] * n + [small_video.frame(5)] * n, defer=True)
    _, A2, _ = ctx.tracker_state(twin[0])
    assert np.array_equal(A2, A0)
    # ... and the commit of one side (a full update) leaves the other side's filters as they were
    ctx.tracker_commit_many(trk, [f1] * n)
    _, A3, _ = ctx.tracker_state(trk[0])
    _, A4, _ = ctx.tracker_state(twin[0])
    assert not np.array_equal(A3, A0) and np.array_equal(A4, A0)
    ctx.tracker_commit_many(twin, [small_video.frame(5)] * n)
    refs = [oracle.Tracker(tabs), oracle.Tracker(tabs)]
    for r, fr, k in zip(refs, (f1, small_video.frame(5)), (0, n)):
        r.start_track(f0, dbox[0])
        assert r.update(fr) == pd[k] and r.get_position() == tuple(bd[k])
    for r, h in zip(refs, (trk[0], twin[0])):
        Ar, Br = r.debug_state()
        _, A, B = ctx.tracker_state(h)
        assert np.array_equal(A, Ar) and np.array_equal(B, Br)
        assert ctx.tracker_position(h) == r.get_position()
    # a clone that is started again lets go of the shared filters
    ctx.tracker_start_many(twin[:1], [f0], dbox[:1])
    _, A5, _ = ctx.tracker_state(twin[0])
    assert np.array_equal(A5, A0)
    ctx.tracker_destroy_many(trk + twin)


def test_pair_mean_dist_and_hac(ctx, oracle):
    rng = np.random.default_rng(5)
    K, T = 9, 60
    centres = rng.normal(0, 1, (K, 128)); centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    sizes = rng.integers(1, 12, T)
    ident = rng.integers(0, K, T)
    rows = []
    for t in range(T):
        x = centres[ident[t]] + 0.05 * rng.normal(0, 1, (sizes[t], 128))
        rows.append(np.round(0.55 * x / np.linalg.norm(x, axis=1, keepdims=True), 5))
    X = np.concatenate(rows)
    rs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    D = ctx.pair_mean_dist(X, rs)
    Dr = oracle.pair_mean_dist(X, rs)
    assert np.allclose(D, Dr, rtol=1e-12, atol=1e-13), np.abs(D - Dr).max()
    labels, log = ctx.cluster_tracks(X, rs, 0.6)
    lr, logr = oracle.hac(Dr, sizes, 0.6)
    assert np.array_equal(labels, lr)
    assert len(log) == len(logr)
    assert np.array_equal(log[:, :2], logr[:, :2])
    # ground truth: tracks of one identity end up together
    for t in range(T):
        assert ident[labels[t]] == ident[t]
    assert len(set(labels.tolist())) == len(set(ident.tolist()))
    # the split form used by several GPUs: row ranges of D stitched together == the whole D, bit for bit; HAC from that D
    from pyannote_video_amd.dist import DistanceShard
    assert np.array_equal(D, D.T)                         # the reference stores matrix[i, j] = matrix[j, i] (clustering.py:111-112)
    for world in (2, 3):
        U = np.zeros_like(D)
        covered = []
        for r in range(world):
            t0, t1 = DistanceShard(r, world).track_range(rs)
            covered.append((t0, t1))
            part = ctx.pair_upper_rows(X, rs, t0, t1)          # the entries j > i of the rows [t0, t1), zeros elsewhere
            assert not part[:t0].any() and not part[t1:].any() and not np.tril(part).any()
            U[t0:t1] = part[t0:t1]
            full = ctx.pair_mean_dist_rows(X, rs, t0, t1)      # the same rows complete (below the diagonal: the mirror), other rows untouched
            assert np.array_equal(full[t0:t1], D[t0:t1]) and not full[:t0].any() and not full[t1:].any()
        assert covered[0][0] == 0 and covered[-1][1] == T and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        assert np.array_equal(U, np.triu(D, 1))
        l2, log2 = ctx.cluster_upper(U, rs, 0.6)               # mirror + agglomeration
        assert np.array_equal(l2, labels) and np.array_equal(log2, log)
        l3, log3 = ctx.cluster_dist(U + U.T, rs, 0.6)
        assert np.array_equal(l3, labels) and np.array_equal(log3, log)
    # the in-memory path: float32 descriptors, table = round(x, 5) in (track, time) order made on the device
    E = X.astype(np.float32)
    perm = rng.permutation(len(E))                               # rows arrive in any order; `order` puts them back
    inv = np.argsort(perm).astype(np.int32)
    Xr = np.round(E.astype(np.float64), 5)
    lf, logf = ctx.cluster_tracks_f32(E[perm], inv, rs, 0.6)
    lw, logw = ctx.cluster_tracks(Xr, rs, 0.6)
    assert np.array_equal(lf, lw) and np.array_equal(logf, logw)  # bit for bit: the same table, the same kernels
    Uf = ctx.pair_upper_rows_f32(E[perm], inv, rs, 3, 41)
    assert Uf.shape == (38, T) and np.array_equal(Uf, np.triu(ctx.pair_mean_dist(Xr, rs), 1)[3:41])


def _tracks(rng, sizes, K=40, noise=0.05):
    centres = rng.normal(0, 1, (K, 128)); centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    ident = rng.integers(0, K, len(sizes))
    rows = []
    for t, n in enumerate(sizes):
        x = centres[ident[t]] + noise * rng.normal(0, 1, (n, 128))
        rows.append(np.round(0.55 * x / np.linalg.norm(x, axis=1, keepdims=True), 5))
    return np.concatenate(rows), np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), ident


def test_pair_mean_dist_matrix_core_kernel_all_block_shapes(ctx, oracle):
    """K10 on the f64 matrix cores: tracks shorter than / equal to / longer than a 16-row block, packed and chunked, identical and
    near-identical rows (the Gram form's cancellation zone), against the oracle's direct double loop"""
    rng = np.random.default_rng(11)
    sizes = np.array([1, 1, 1, 15, 16, 17, 2, 3, 31, 32, 33, 1, 5, 5, 5, 1, 64, 7, 9, 250, 1, 1], np.int64)
    X, rs, _ = _tracks(rng, sizes)
    X[rs[3] + 1] = X[rs[3]]                                  # identical rows inside a track and across tracks
    X[rs[5]] = X[rs[4]]
    X[rs[8] + 2] = X[rs[8] + 1] + 1e-5
    D = ctx.pair_mean_dist(X, rs)
    Dr = oracle.pair_mean_dist(X, rs)
    assert np.allclose(D, Dr, rtol=1e-12, atol=1e-13), np.abs(D - Dr).max()
    assert not D.diagonal().any()
    # row ranges stitched == whole, bit for bit, wherever the cuts fall
    assert np.array_equal(D, D.T)
    for cuts in ([0, 4, 9, 22], [0, 1, 20, 22], [0, 10, 22]):
        Ds = np.zeros_like(D)
        for t0, t1 in zip(cuts, cuts[1:]):
            Ds[t0:t1] = ctx.pair_upper_rows(X, rs, t0, t1)[t0:t1]
        assert np.array_equal(Ds, np.triu(D, 1))           # upper-triangle shares; mirrored == the whole matrix
        assert np.array_equal(Ds + Ds.T, D)
        Df = np.zeros_like(D)
        for t0, t1 in zip(cuts, cuts[1:]):
            Df[t0:t1] = ctx.pair_mean_dist_rows(X, rs, t0, t1)[t0:t1]      # complete rows stitch into the whole matrix
        assert np.array_equal(Df, D)
    # cosine distance (north_star's metric): mean of 1 - cos over the block
    Dc = ctx.pair_mean_dist(X, rs, metric=1)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    full = 1.0 - Xn @ Xn.T
    want = np.array([[full[rs[i]:rs[i + 1], rs[j]:rs[j + 1]].mean() if i != j else 0.0 for j in range(len(sizes))] for i in range(len(sizes))])
    assert np.allclose(Dc, want, rtol=1e-10, atol=1e-12)


def test_hac_persistent_kernel_3000_tracks_equals_oracle(ctx, oracle):
    """K11 as one persistent workgroup: T = 3000 tracks (1-6 rows), labels and merge order == the CPU oracle"""
    rng = np.random.default_rng(12)
    T = 3000
    sizes = rng.integers(1, 7, T)
    X, rs, ident = _tracks(rng, sizes, K=150)
    labels, log = ctx.cluster_tracks(X, rs, 0.6)
    Dr = oracle.pair_mean_dist(X, rs)
    lr, logr = oracle.hac(Dr, sizes, 0.6)
    assert np.array_equal(labels, lr)
    assert len(log) == len(logr) and np.array_equal(log[:, :2], logr[:, :2])
    assert np.allclose(log[:, 2], logr[:, 2], rtol=1e-11)
    for t in range(T):
        assert ident[labels[t]] == ident[t]


@pytest.mark.parametrize("T", [40, 700, 3200])
def test_hac_tie_order_equals_oracle(ctx, oracle, T):
    """K11 on distance matrices made of a handful of values: almost every minimum is tied, so the merge order is decided by the
    first-minimum-in-row-major-order rule alone (in the wave reductions, in the re-scans, in the cached row minima); T = 40 / 700 / 3200
    run the three instantiations of the persistent kernel (<= 1024, <= 3072, <= 10 240 tracks)"""
    rng = np.random.default_rng(100 + T)
    lv = np.array([0.125, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5])
    D = lv[rng.integers(0, len(lv), (T, T))]
    D = np.triu(D, 1); D = D + D.T
    sizes = rng.integers(1, 5, T)
    rs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    labels, log = ctx.cluster_dist(D, rs, 0.6)
    lr, logr = oracle.hac(D, sizes, 0.6)
    assert len(log) == len(logr) and T // 2 < len(log) < T - 1
    assert np.array_equal(log[:, :2], logr[:, :2])
    assert np.array_equal(log[:, 2:], logr[:, 2:])           # distances and sizes: the same operations in the same order
    assert np.array_equal(labels, lr)


def test_detector_with_hundreds_of_candidates_at_the_threshold(ctx, oracle, small_video):
    """the synthetic detector is fit to the renderer, so at the shipped threshold few window scores sit near it.
    Here `adjust_threshold` is lowered until ~20 000 windows of one frame pass (bisection on the CPU oracle) -- the threshold then lies in
    the dense part of the score distribution -- and the raw candidates (score bits, filter, level, position, box) and the final boxes
    must still equal the oracle's at that threshold and 1e-3 to either side of it; the counts differ by >= 100 between the two outer
    thresholds, i.e. at least that many windows score within 1e-3 of the middle one: every one of them decided the same way on both sides."""
    det = _detector(oracle)
    f = small_video.frame(5)
    lo, hi = -6.0, 0.0                                   # adjust: candidates(lo) > target > candidates(hi)
    n_lo = len(det.detect_raw(f, 1, lo))
    assert n_lo > 20000, n_lo
    for _ in range(12):
        mid = 0.5 * (lo + hi)
        if len(det.detect_raw(f, 1, mid)) > 20000:
            lo = mid
        else:
            hi = mid
    A = float(np.float32(lo))
    counts = []
    for adj in (A - 1e-3, A, A + 1e-3):
        raw_c = det.detect_raw(f, 1, adj)
        raw_g = ctx.detect_raw(f, 1, adj)
        assert 1000 < len(raw_c) < 65000
        assert len(raw_g) == len(raw_c), (adj, len(raw_g), len(raw_c))
        assert raw_g == raw_c
        counts.append(len(raw_c))
        boxes_g, scores_g = ctx.detect(f, 1, adj)
        fin_c = det.detect(f, 1, adj)
        assert boxes_g == [d[5] for d in fin_c]
        assert np.array_equal(scores_g, np.array([d[0] for d in fin_c], np.float32))
    assert counts[0] - counts[2] >= 100, counts          # windows within 1e-3 of the middle threshold


def test_detector_border_windows_after_tracker_work(ctx, oracle, small_video):
    """Round-4 finding of the test above: the tracker used the detector's feature scratch for its chip features and overwrote the ZERO
    BORDER of the level-0 feature maps, so a detector call that followed tracker work (every batch of a pipeline run but the first)
    scored the windows reaching into the border on stale tracker features -- invisible at the shipped threshold unless a face touches
    the frame's edge.  Detector -> tracker starts + updates -> detector with the threshold in the dense part of the score distribution:
    every raw candidate (20 000 of them, the border rows included) must still equal the oracle's."""
    det = _detector(oracle)
    f0, f1, f = small_video.frame(0), small_video.frame(1), small_video.frame(5)
    boxes = ctx.detect(f0, 1)[0]
    dbox = [tuple(float(v) for v in b) for b in boxes]
    trk = ctx.tracker_create_many(len(dbox))
    ctx.tracker_start_many(trk, [f0] * len(dbox), dbox)
    ctx.tracker_update_many(trk, [f1] * len(dbox))
    adj = -0.80712890625
    raw_c = det.detect_raw(f, 1, adj)
    raw_g = ctx.detect_raw(f, 1, adj)
    assert len(raw_c) > 15000 and min(r[3] for r in raw_c) == 5           # windows of the first scanned row are among them
    assert raw_g == raw_c
    ctx.tracker_destroy_many(trk)


def test_cluster_ten_thousand_tracks_equals_the_oracle_fixture(ctx):
    """BASELINE.json configs[4]'s clustering stressor at FULL size (T = 10 000 tracks x 10 rows, N = 1e5; tools/c5_cluster.py's generator
    and seed) against the CPU oracle's frozen result (tests/golden/c5_cluster_T10000.npz, made by tests/golden/make_c5_cluster.py in a
    minute of CPU): labels, every merge in order, merge distances -- through pvf_cluster_tracks and through the float32 in-memory path."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import importlib
    c5 = importlib.import_module("c5_cluster")
    g = np.load(os.path.join(root, "tests", "golden", "c5_cluster_T10000.npz"))
    X, rs, ident = c5.make(10000, 10)
    labels, log = ctx.cluster_tracks(X, rs, 0.6)
    assert np.array_equal(labels, g["labels"])
    assert len(log) == len(g["merge_pairs"]) == 9500 and len(set(labels.tolist())) == 500
    assert np.array_equal(np.asarray(log)[:, :2].astype(np.int32), g["merge_pairs"])
    assert np.abs(np.asarray(log)[:, 2] - g["merge_dist"]).max() <= 1e-12
    lf, _ = ctx.cluster_tracks_f32(X.astype(np.float32), None, rs, 0.6)
    assert np.array_equal(lf, g["labels"])
