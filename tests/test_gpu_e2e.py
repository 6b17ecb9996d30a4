"""GPU: the whole flow (detect -> track fwd/bwd -> landmarks -> embed -> cluster) through the product pipeline against the
sequential CPU oracle flow on the same synthetic clip.  Track boxes / ids / statuses, landmarks and cluster labels are
compared exactly; embeddings within L2 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipeline_matches_oracle_flow(ctx, oracle, small_video, model_paths):
    from pyannote_video_amd import models, pipeline
    from oracle import ref_flow
    v = small_video
    frames_np = [v.frame(i) for i in range(v.n_frames)]
    times = [v.timestamp(i) for i in range(v.n_frames)]
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_batch_size=5)
    res = pipe.run([ctx.upload(f) for f in frames_np], times, v.frame_rate, v.shots())

    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    tabs = models.dsst_tables()
    ref_tracks = ref_flow.track_video(frames_np, times, v.shots(), det, lambda: oracle.Tracker(tabs), v.frame_rate,
                                      min_conf=10., ratio=0.5, max_gap=1.0)
    assert len(ref_tracks) >= v.faces * v.n_shots
    assert res["tracks"] == ref_tracks
    lines = ref_flow.track_text(ref_tracks)
    assert lines == [l.rstrip("\n") for i, tr in enumerate(res["tracks"]) for l in __import__("pyannote_video_amd").formats.track_lines(i, tr)]
    got_pts = []
    lm, em = ref_flow.extract(lines, frames_np, times, lambda f, b: got_pts.append(sp(f, b)) or got_pts[-1], emb)
    assert len(got_pts) == len(res["landmarks"]) > 0
    assert np.array_equal(np.stack(got_pts), res["landmarks"])
    ref_e = np.array([[float(x) for x in line.split()[2:]] for line in em]).reshape(-1, 128)
    assert np.linalg.norm(ref_e - res["embeddings"], axis=1).max() <= 1e-4 + 128 ** 0.5 * 5e-6   # text rounding of the reference side
    assert [int(l.split()[1]) for l in em] == res["face_id"].tolist()
    assert ref_flow.cluster(em, 0.6) == res["labels"]


def test_pipeline_with_sparse_detection_matches_oracle_flow(ctx, oracle, small_video, model_paths):
    """detection every 3rd frame: trackers have to live between detections, i.e. the deferred first updates of the bulk path are
    committed and followed by on-demand updates; tracks must still equal the sequential oracle flow"""
    from pyannote_video_amd import models, pipeline
    from oracle import ref_flow
    v = small_video
    frames_np = [v.frame(i) for i in range(v.n_frames)]
    times = [v.timestamp(i) for i in range(v.n_frames)]
    every = 3.0 / v.frame_rate
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_every=every, detect_batch_size=4)
    res = pipe.run([ctx.upload(f) for f in frames_np], times, v.frame_rate, v.shots(), cluster=False)
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    tabs = models.dsst_tables()
    ref_tracks = ref_flow.track_video(frames_np, times, v.shots(), det, lambda: oracle.Tracker(tabs), v.frame_rate,
                                      detect_every=every, min_conf=10., ratio=0.5, max_gap=1.0)
    assert any("forward" in st or "backward" in st for tr in ref_tracks for _, _, st in tr)
    assert res["tracks"] == ref_tracks


def test_reference_api_surface(ctx, small_video, model_paths):
    """Face / FaceTracking / FaceClustering used the way scripts/pyannote-face.py uses them (:247-267, :281-311)"""
    from pyannote_video_amd import Face, FaceTracking, FaceClustering
    from pyannote_video_amd._core import Segment
    from pyannote_video_amd import shim
    v = small_video
    tracking = FaceTracking(track_min_overlap_ratio=0.5, track_max_gap=1.0, ctx=ctx)
    shots = [Segment(a, b) for a, b in v.shots()]
    tracks = list(tracking(v, shots))
    assert len(tracks) >= v.faces * v.n_shots
    face = Face(landmarks=model_paths[0], embedding=model_paths[1], ctx=ctx)
    rgb = v.frame(0)
    faces = list(face.iterfaces(rgb))
    assert len(faces) == v.faces and all(isinstance(f, shim.rectangle) for f in faces)
    lms = face.get_landmarks(rgb, faces[0])
    assert len(lms.parts()) == 68
    e = list(face.get_embedding(rgb, lms))
    assert len(e) == 128
    triples = list(face(rgb, return_landmarks=True, return_embedding=True))
    assert len(triples) == v.faces and np.allclose(list(triples[0][2]), e)
    trk = shim.correlation_tracker(ctx)
    trk.start_track(rgb, shim.drectangle(*[float(x) for x in faces[0].as_tuple()]))
    conf = trk.update(v.frame(1))
    pos = trk.get_position()
    assert conf > 5 and pos.intersect(shim.drectangle(*faces[0].as_tuple())).area() > 0
    ctx.unstage_all()


def test_device_resize_equals_cv2_restatement(ctx, oracle, small_video):
    """pvf_frame_resize (OpenCV's 8-bit bilinear on the device, reference video.py:402-403) == the oracle's restatement, byte for byte"""
    f = small_video.frame(0)
    dev = ctx.upload(f)
    for (w, h) in [(320, 180), (333, 187), (640, 360), (500, 300), (97, 55)]:
        got = ctx.pyramid_level(ctx.resize(dev, w, h), 0, 0)          # level 0 without upsampling = the frame's bytes
        assert got.shape == (h, w, 3) and np.array_equal(got, oracle.cv_resize(f, w, h)), (w, h)


def test_ingest_ring_frames_equal_direct_uploads(ctx, small_video):
    """pinned ring + asynchronous uploads on the copy stream: the frames that arrive are the frames that were pushed, in every slot
    reuse pattern (ring shorter than the clip), and kernels see them only after the copy (lazy wait on first use)"""
    v = small_video
    frames = [v.frame(i) for i in range(v.n_frames)]
    ring = ctx.ingest_ring(v.size[1], v.size[0], depth=3)
    dev = [ring.push(f) for f in frames]                           # 12 frames through 3 slots
    want = ctx.detect_batch([ctx.upload(f) for f in frames], 1)
    got = ctx.detect_batch(dev, 1)
    assert [g[0] for g in got] == [w[0] for w in want] and all(np.array_equal(g[1], w[1]) for g, w in zip(got, want))
    for i in (0, 5, 11):
        assert np.array_equal(ctx.pyramid_level(dev[i], 0, 0), frames[i])
    # a decoder writing straight into the slot
    s = ring.slot()
    s[...] = frames[3]
    assert np.array_equal(ctx.pyramid_level(ring.submit(), 0, 0), frames[3])
    ring.close()


def test_min_size_detection_on_downscaled_frames_matches_oracle_flow(ctx, oracle, small_video, model_paths):
    """--min-size (tracking.py:389-400): detect + track on frames resized so that the smallest wanted face is 36 px, boxes normalised by
    the resized size, landmarks / embeddings on the native frames (pyannote-face.py:275-277)"""
    from pyannote_video_amd import models, pipeline
    from oracle import ref_flow
    v = small_video
    frames_np = [v.frame(i) for i in range(v.n_frames)]
    times = [v.timestamp(i) for i in range(v.n_frames)]
    min_size = 0.14                                               # 36 / (0.14 * 360) = 0.714 -> 457 x 257
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_min_size=min_size, detect_batch_size=4)
    res = pipe.run([ctx.upload(f) for f in frames_np], times, v.frame_rate, v.shots())
    ratio = min(1.0, 36 / (min_size * v.size[1]))
    tw, th = int(v.size[0] * ratio), int(v.size[1] * ratio)
    assert (tw, th) == (457, 257)
    small = [oracle.cv_resize(f, tw, th) for f in frames_np]
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    tabs = models.dsst_tables()
    ref_tracks = ref_flow.track_video(small, times, v.shots(), det, lambda: oracle.Tracker(tabs), v.frame_rate, min_conf=10., ratio=0.5, max_gap=1.0)
    assert len(ref_tracks) >= 4
    assert res["tracks"] == ref_tracks
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    got_pts = []
    lm, em = ref_flow.extract(ref_flow.track_text(ref_tracks), frames_np, times, lambda f, b: got_pts.append(sp(f, b)) or got_pts[-1], emb)
    assert len(got_pts) == len(res["landmarks"]) > 0 and np.array_equal(np.stack(got_pts), res["landmarks"])
    assert ref_flow.cluster(em, 0.6) == res["labels"]
    # the drop-in class does the same from a source that yields native frames
    from pyannote_video_amd import FaceTracking
    from pyannote_video_amd._core import Segment
    tracking = FaceTracking(detect_min_size=min_size, track_min_overlap_ratio=0.5, track_max_gap=1.0, ctx=ctx)
    assert list(tracking(v, [Segment(a, b) for a, b in v.shots()])) == ref_tracks
