"""GPU parity at the configuration bench.py times (BASELINE.json configs[1]): 1920 x 1080, upsample 1 (20 pyramid levels), batches of
32 frames, the FULL landmark model (15 cascades x 500 trees x 500 pixels), 4096-tracker bulk calls -- plus the detector alone at
configs[3] (1280 x 720) and configs[4] (3840 x 2160).  Everything is compared with the CPU oracle on the same bytes:
raw candidates, NMS boxes and scores, landmarks, chips (bit-exact), embeddings (L2 <= 1e-4), tracker PSR and positions (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_models(full_model_paths):
    return full_model_paths


@pytest.fixture(scope="module")
def clip1080():
    from pyannote_video_amd import synth
    # the bench generator at its own size; 4 frames that straddle the first shot cut (frames 248..251 of the 1000-frame clip layout)
    v = synth.SyntheticVideo(width=1920, height=1080, n_frames=8, n_shots=2, faces=8, seed=20260925)
    return v, [v.frame(i) for i in (2, 3, 4, 5)]


def _oracle_detector(oracle):
    from pyannote_video_amd import models
    oracle.lib().pvo_set_threads(64)
    return oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))


def test_detector_1080p_batches_bit_exact(ctx_full, oracle, clip1080):
    _, frames = clip1080
    det = _oracle_detector(oracle)
    want = [det.detect(f, 1) for f in frames]
    assert all(len(w) >= 6 for w in want)                       # ~8 faces per frame fire
    dev = [ctx_full.upload(f) for f in frames]
    # raw candidates of one frame (single-frame plan) == oracle, record for record
    raw = ctx_full.detect_raw(dev[0], 1)
    ref = det.detect_raw(frames[0], 1)
    assert len(raw) == len(ref) > 0
    assert raw == [(r[0], r[1], r[2], r[3], r[4], tuple(r[5])) for r in ref]
    # batches of 32, 64 and 128 (the four frames repeated; 128 is the bench's batch) through the pipelined entry: every copy equals the oracle
    for batch in (32, 64, 128):
        res = ctx_full.detect_many([dev[i % 4] for i in range(2 * batch)], batch, 1)
        for i, (boxes, scores) in enumerate(res):
            w = want[i % 4]
            assert boxes == [tuple(d[5]) for d in w]
            assert np.array_equal(scores, np.array([d[0] for d in w], np.float32))


def test_full_landmark_model_chips_and_embeddings_1080p(ctx_full, oracle, clip1080, full_models):
    from pyannote_video_amd import models
    _, frames = clip1080
    det = _oracle_detector(oracle)
    sp = oracle.ShapePredictor(models.load_container(full_models[0]))
    emb = oracle.Embedder(models.load_container(full_models[1]))
    fr, boxes = [], []
    for f in frames[:2]:
        for d in det.detect(f, 1):
            fr.append(f); boxes.append(tuple(d[5]))
    assert len(boxes) >= 12
    pts = ctx_full.landmarks(fr, boxes)
    want = np.stack([sp(f, b) for f, b in zip(fr, boxes)])
    assert np.array_equal(pts, want)                           # 15 x 500 trees, 500 feature pixels: bit-exact integer points
    chips = ctx_full.face_chips(fr, pts)
    assert np.array_equal(chips, np.stack([emb.chip(f, p) for f, p in zip(fr, want)]))
    e = ctx_full.embed(fr, pts)
    ref = np.stack([emb(f, p) for f, p in zip(fr, want)])
    assert np.linalg.norm(e.astype(np.float64) - ref.astype(np.float64), axis=1).max() <= 1e-4


def test_trackers_1080p_bulk_bit_exact(ctx_full, oracle, clip1080):
    from pyannote_video_amd import models
    _, frames = clip1080
    det = _oracle_detector(oracle)
    boxes = [tuple(float(v) for v in d[5]) for d in det.detect(frames[0], 1)]
    n = len(boxes)
    tabs = models.dsst_tables()
    ref = []
    for b in boxes:
        t = oracle.Tracker(tabs)
        t.start_track(frames[0], b)
        ref.append(t)
    want = [(t.update(frames[1]), t.get_position()) for t in ref]
    want2 = [(t.update(frames[2]), t.get_position()) for t in ref]
    hs = ctx_full.tracker_create_many(n)
    dev = [ctx_full.upload(f) for f in frames[:3]]
    ctx_full.tracker_start_many(hs, [dev[0]] * n, boxes)
    psr, pos = ctx_full.tracker_update_many(hs, [dev[1]] * n, defer=True)       # the bulk path: deferred, then committed
    assert psr.tolist() == [w[0] for w in want]
    assert [tuple(p) for p in pos] == [w[1] for w in want]
    ctx_full.tracker_commit_many(hs, [dev[1]] * n)
    psr, pos = ctx_full.tracker_update_many(hs, [dev[2]] * n)
    assert psr.tolist() == [w[0] for w in want2]
    assert [tuple(p) for p in pos] == [w[1] for w in want2]
    ctx_full.tracker_destroy_many(hs)


@pytest.mark.parametrize("size", [(1280, 720), (3840, 2160)])
def test_detector_other_configs_single_frame(ctx_full, oracle, size):
    """configs[3] (720p clips) and configs[4] (4K crowd): raw candidates and boxes of one frame == oracle"""
    from pyannote_video_amd import synth
    w, h = size
    v = synth.SyntheticVideo(width=w, height=h, n_frames=2, n_shots=1, faces=8 if w < 3000 else 40, seed=20260925 + w)
    f = v.frame(1)
    det = _oracle_detector(oracle)
    ref = det.detect_raw(f, 1)
    raw = ctx_full.detect_raw(f, 1, cap=1 << 17)
    assert len(raw) == len(ref) > 0
    assert raw == [(r[0], r[1], r[2], r[3], r[4], tuple(r[5])) for r in ref]
    boxes, scores = ctx_full.detect(f, 1)
    want = det.detect(f, 1)
    assert boxes == [tuple(d[5]) for d in want] and len(boxes) >= (6 if w < 3000 else 25)
    assert np.array_equal(scores, np.array([d[0] for d in want], np.float32))


def test_crowded_frame_overflows_the_default_slots_and_dense_nms(ctx_full, oracle):
    """A 1080p frame with 96 faces: more detections than the 64 result slots per frame `detect_many` starts with (the call repeats itself
    with room for every detection), and -- with the threshold lowered by 0.5 -- thousands of raw candidates (beyond the 512 per frame
    copied back ahead: the on-demand fetch) whose overlapping boxes the greedy suppression has to walk in the canonical order.  Boxes,
    scores and their order == the CPU oracle, also in batches where crowded and ordinary frames alternate."""
    from pyannote_video_amd import synth
    crowd = synth.SyntheticVideo(width=1920, height=1080, n_frames=2, n_shots=1, faces=96, min_face=60, max_face=120, seed=77)
    plain = synth.SyntheticVideo(width=1920, height=1080, n_frames=2, n_shots=1, faces=8, seed=20260925)
    fc, fp = crowd.frame(1), plain.frame(0)
    det = _oracle_detector(oracle)
    want_c, want_p = det.detect(fc, 1), det.detect(fp, 1)
    assert len(want_c) > 64 and len(want_p) >= 6
    dev = [ctx_full.upload(fc), ctx_full.upload(fp)]
    res = ctx_full.detect_many([dev[i % 2] for i in range(12)], 4, 1)              # cap = 64 slots -> repeated with 512
    for i, (boxes, scores) in enumerate(res):
        w = want_c if i % 2 == 0 else want_p
        assert boxes == [tuple(d[5]) for d in w]
        assert np.array_equal(scores, np.array([d[0] for d in w], np.float32))
    # lowered threshold: a dense candidate cloud, more than the 8192 candidate slots a frame starts with (they grow on demand)
    raw = det.detect_raw(fc, 1, -0.5)
    assert len(raw) > 8192
    low = det.detect(fc, 1, -0.5)
    out, sc, cnt = ctx_full.detect_many([dev[0], dev[1], dev[0]], 2, 1, adjust_threshold=-0.5, arrays=True)
    assert cnt[0] == cnt[2] == len(low) > len(want_c)
    assert out[0, :cnt[0]].tolist() == [list(d[5]) for d in low] and np.array_equal(out[2, :cnt[2]], out[0, :cnt[0]])
    assert np.array_equal(sc[0, :cnt[0]], np.array([d[0] for d in low], np.float32))
