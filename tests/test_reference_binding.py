"""The reference's own CLI bodies -- `track` and `extract` of /root/reference/scripts/pyannote-face.py:239-314, with the reference's
FaceTracking / TrackingByDetection / Face classes and the real munkres package, all executed verbatim (tests/refhost.py) -- against:

  CPU   (a) the committed fixtures tests/golden/reference_cli_small*/  (they are that run's output, `dlib` = the CPU oracle);
        (b) oracle/ref_flow.py, the sequential restatement the other parity tests compare with  => ref_flow is pinned to the
            reference's executed code, not to a reading of it;
  GPU   the product (FacePipeline through the C ABI) must write the same track.txt / landmarks.txt byte for byte and embeddings
        within 1e-4 (+ the 5-decimal rounding of the file); where /root/reference exists next to a GPU, the reference CLI also
        runs directly on pyannote_video_amd.shim (executed).
"""
import os
import sys
import numpy as np
import pytest
import refhost

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_reference_golden as mrg   # noqa: E402

needs_reference = pytest.mark.skipif(not (refhost.have_reference() and refhost.have_munkres()), reason="/root/reference not on this machine")
GOLD = {0.0: os.path.join(HERE, "golden", "reference_cli_small"), 0.12: os.path.join(HERE, "golden", "reference_cli_small_every3")}


def _lines(path):
    with open(path) as f:
        return f.read().splitlines()


def _clip():
    from pyannote_video_amd import synth
    v = synth.SyntheticVideo(**mrg.CLIP)
    return v, [v.frame(i) for i in range(v.n_frames)], [v.timestamp(i) for i in range(v.n_frames)]


@needs_reference
def test_reference_cli_reproduces_the_committed_fixtures(tmp_path, model_paths):
    import oracle_dlib
    from pyannote_video_amd import models
    oracle_dlib.configure(models.load_container(models.DEFAULT_DETECTOR), models.dsst_tables())
    paths = mrg.run_reference_cli(str(tmp_path), oracle_dlib, model_paths[0], model_paths[1])
    for p, name in zip(paths, ("track.txt", "landmarks.txt", "embedding.txt")):
        assert _lines(p) == _lines(os.path.join(GOLD[0.0], name)), name


@pytest.mark.parametrize("every", [0.0, 0.12])
def test_sequential_restatement_equals_reference_cli_output(every, oracle, model_paths):
    """oracle/ref_flow.py (what the GPU parity tests compare with) == files the reference's own code wrote"""
    from oracle import ref_flow
    from pyannote_video_amd import models, pipeline
    v, frames, times = _clip()
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    tabs = models.dsst_tables()
    tracks = ref_flow.track_video(frames, times, v.shots(), det, lambda: oracle.Tracker(tabs), v.frame_rate, detect_every=every,
                                  min_conf=pipeline.CLI_MIN_CONFIDENCE, ratio=pipeline.CLI_MIN_OVERLAP_RATIO, max_gap=pipeline.CLI_MAX_GAP)
    assert ref_flow.track_text(tracks) == _lines(os.path.join(GOLD[every], "track.txt"))
    lm, em = ref_flow.extract(_lines(os.path.join(GOLD[every], "track.txt")), frames, times, sp, emb)
    assert lm == _lines(os.path.join(GOLD[every], "landmarks.txt"))
    assert em == _lines(os.path.join(GOLD[every], "embedding.txt"))


def _canon(lines):
    """rows of one timestamp in track order.  Their order in landmarks.txt / embedding.txt comes from pandas' unstable sort of the
    track table (formats.pandas_sort_order), i.e. from numpy's sort kernel for the CPU at hand; the fixtures were written on the
    build container, the product runs on the GPU box, so the comparison is made independent of that one degree of freedom."""
    return sorted(lines, key=lambda l: (float(l.split()[0]), int(l.split()[1])))


def _product_files(ctx, model_paths, every):
    from pyannote_video_amd import pipeline, formats
    v, frames, times = _clip()
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_every=every)
    res = pipe.run([ctx.upload(f) for f in frames], times, v.frame_rate, v.shots())
    w, h = v.frame_size
    track = [l.rstrip("\n") for i, tr in enumerate(res["tracks"]) for l in formats.track_lines(i, tr)]
    lm = [formats.landmark_line(T, int(i), p, w, h).rstrip("\n") for T, i, p in zip(res["face_T"], res["face_id"], res["landmarks"])]
    return track, lm, res


@pytest.mark.gpu
@pytest.mark.parametrize("every", [0.0, 0.12])
def test_product_writes_what_the_reference_cli_wrote(every, ctx, model_paths):
    track, lm, res = _product_files(ctx, model_paths, every)
    assert track == _lines(os.path.join(GOLD[every], "track.txt"))                 # byte for byte: ids, 3-decimal boxes, status strings
    assert _canon(lm) == _canon(_lines(os.path.join(GOLD[every], "landmarks.txt")))   # byte for byte: 68 integer points per face
    gold = np.array([[float(x) for x in l.split()] for l in _canon(_lines(os.path.join(GOLD[every], "embedding.txt")))])
    order = np.lexsort((res["face_id"], res["face_T"]))
    assert np.array_equal(gold[:, 0], res["face_T"][order]) and np.array_equal(gold[:, 1].astype(int), res["face_id"][order])
    assert np.linalg.norm(gold[:, 2:] - res["embeddings"][order].astype(np.float64), axis=1).max() <= 1e-4 + 128 ** 0.5 * 0.5e-5
    # same cluster labels as an agglomeration of the reference's own embedding file
    from oracle import ref_flow
    assert res["labels"] == ref_flow.cluster(_lines(os.path.join(GOLD[every], "embedding.txt")), 0.6)


@pytest.mark.gpu
@needs_reference
def test_reference_cli_runs_on_the_hip_shim(tmp_path, ctx, model_paths):
    """INTEGRATION.md section 1 executed: the reference's track()/extract() with `import dlib` re-pointed at the HIP library"""
    from pyannote_video_amd import shim, runtime
    runtime._default = ctx
    try:
        paths = mrg.run_reference_cli(str(tmp_path), shim, model_paths[0], model_paths[1])
    finally:
        runtime._default = None
    assert _lines(paths[0]) == _lines(os.path.join(GOLD[0.0], "track.txt"))
    assert _canon(_lines(paths[1])) == _canon(_lines(os.path.join(GOLD[0.0], "landmarks.txt")))
    a = np.array([[float(x) for x in l.split()] for l in _canon(_lines(paths[2]))])
    b = np.array([[float(x) for x in l.split()] for l in _canon(_lines(os.path.join(GOLD[0.0], "embedding.txt")))])
    assert np.abs(a - b).max() <= 1e-4 + 1e-5


# ---- the reference's call sequence into dlib, recorded where /root/reference exists, replayed where the GPU is ----------------------
def _trace(every):
    import json
    with open(os.path.join(GOLD[every], "dlib_trace.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("every", [0.0, 0.12])
def test_recorded_reference_calls_replay_on_the_oracle_dlib(every, oracle, model_paths):
    """the trace is self-consistent (replayed against the look-alike it was recorded from it reproduces every return value) and holds
    what the reference does: one detector call per detection frame (face.py:66 via tracking.py:426), one tracker object per (detection,
    pass) started once (tracking.py:250-251) and updated frame by frame (:203), positions read after updates (:231), one landmark and
    one descriptor call per extracted face (pyannote-face.py:296-297)"""
    import dlib_trace
    import oracle_dlib
    from pyannote_video_amd import models
    oracle_dlib.configure(models.load_container(models.DEFAULT_DETECTOR), models.dsst_tables())
    t = _trace(every)
    assert t["meta"]["clip"] == mrg.CLIP and t["meta"]["every"] == every
    v, frames, _ = _clip()
    n = dlib_trace.replay(oracle_dlib, t["calls"], frames, model_paths[0], model_paths[1], embed_tol=0.0)
    n_faces = len(_lines(os.path.join(GOLD[every], "landmarks.txt")))
    assert n["detector.call"] == (12 if every == 0.0 else 4)
    assert n["shape_predictor.call"] == n["face_recognition.call"] == n_faces
    assert n["tracker.new"] == n["tracker.start_track"] == 2 * sum(len(c[3]) for c in t["calls"] if c[0] == "detector.call")
    # every tracker is updated at least once except those started on the last frame of their pass (3 faces x 2 shots x 2 passes)
    assert n["tracker.update"] >= n["tracker.new"] - 12 and n["tracker.get_position"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("every", [0.0, 0.12])
def test_recorded_reference_calls_replay_on_the_hip_shim(every, ctx, model_paths):
    """INTEGRATION.md section 1 without the reference files: the exact dlib calls the reference's track() + extract() made (recorded in
    the build container, tests/golden/make_reference_golden.py) go through pyannote_video_amd.shim -> C ABI -> HIP kernels one by one --
    per-object trackers, single-frame detector calls, landmark objects handed on to the embedder -- and every return value equals what
    the CPU oracle returned to the reference: boxes, tracker confidences and positions, 68 points bit for bit, descriptors within 1e-4"""
    import dlib_trace
    from pyannote_video_amd import shim, runtime
    v, frames, _ = _clip()
    old = runtime._default
    runtime._default = ctx
    try:
        n = dlib_trace.replay(shim, _trace(every)["calls"], frames, model_paths[0], model_paths[1], embed_tol=1e-4)
    finally:
        runtime._default = old
        ctx.unstage_all()
    assert n["tracker.update"] > 20 and n["face_recognition.call"] > 20
