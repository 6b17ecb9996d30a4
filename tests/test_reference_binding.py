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
