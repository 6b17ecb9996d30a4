"""Pins to the REAL third-party code this container holds, and to the reference's own Python executed verbatim
(tests/refhost.py).  CPU only; skipped where /root/reference or the munkres source is absent (GPU box).

  munkres 1.1.4 (reference setup.py:51, tracking.py:35,121,172): its source is on disk; the product's pvf_munkres / pvf_associate
  and the oracle's pvo_munkres must return the package's own assignment -- ties included (SURVEY.md A.6 "port literally").
  pyannote/video/tracking.py: imports only numpy, networkx, munkres, dlib -> runs here against a scripted `dlib`.
"""
import types
import numpy as np
import pytest
import refhost
from pyannote_video_amd import _lib
from pyannote_video_amd.tracking_by_detection import TrackingByDetection, ObjectTrackers
from test_host_logic import ScriptTracker, ScriptFrame, ScriptVideo, scenario

needs_munkres = pytest.mark.skipif(not refhost.have_munkres(), reason="munkres source not on this machine")
needs_reference = pytest.mark.skipif(not (refhost.have_reference() and refhost.have_munkres()), reason="/root/reference not on this machine")


def _real(cost):
    m = refhost.real_munkres().Munkres()
    return sorted(m.compute([list(map(float, row)) for row in cost]))


def _tie_heavy(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        return rng.integers(0, 3, (n, n)).astype(np.float64)                  # few distinct values
    if kind == 1:
        return np.round(rng.uniform(0, 1, (n, n)), 1)                          # one decimal
    if kind == 2:                                                              # _associate-shaped: max - overlap, zero padded
        nt, nd = int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))
        ov = np.zeros((n, n))
        ov[:nt, :nd] = rng.choice([0.0, 0.0, 900.0, 1600.0, 2500.0], (nt, nd))
        return ov.max() - ov
    return rng.uniform(0, 100, (n, n))                                         # distinct


@needs_munkres
def test_munkres_equals_the_real_package_ties_included(oracle):
    rng = np.random.default_rng(1)
    for _ in range(3000):
        n = int(rng.integers(1, 9))
        cost = _tie_heavy(rng, n)
        want = _real(cost)
        assert sorted(_lib.munkres(cost)) == want, cost
        assert sorted(oracle.munkres(cost)) == want, cost
    for n in (12, 20, 33):                                                     # larger, many shift steps
        for _ in range(20):
            cost = _tie_heavy(rng, n)
            want = _real(cost)
            assert sorted(_lib.munkres(cost)) == want
            assert sorted(oracle.munkres(cost)) == want


def _scripted_dlib():
    """a `dlib` with the scripted tracker of test_host_logic and real rectangle semantics"""
    import oracle_dlib
    mod = types.ModuleType("dlib")
    mod.drectangle = oracle_dlib.drectangle
    mod.rectangle = oracle_dlib.rectangle

    class correlation_tracker(ScriptTracker):
        def start_track(self, frame, box):
            self.box = (box.left(), box.top(), box.right(), box.bottom())

        def get_position(self):
            return oracle_dlib.drectangle(*self.box)
    mod.correlation_tracker = correlation_tracker
    mod.get_frontal_face_detector = lambda: None
    mod.shape_predictor = mod.face_recognition_model_v1 = lambda path: None
    return mod


@needs_reference
def test_associate_equals_reference_method():
    """the reference's own TrackingByDetection._associate (tracking.py:136-182, real munkres) vs pvf_associate"""
    rng = np.random.default_rng(2)
    with refhost.reference_modules(_scripted_dlib()) as ref:
        tbd = ref.tracking.TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5)
        dl = ref.tracking.dlib

        class T(object):
            def __init__(self, box): self.box = box
            def get_position(self): return dl.drectangle(*self.box)
        for _ in range(1500):
            nt, nd = int(rng.integers(1, 7)), int(rng.integers(1, 7))
            # boxes on a coarse grid: equal overlaps (ties) are common
            def boxes(n):
                xy = rng.integers(0, 4, (n, 2)) * 20
                s = rng.choice([40, 60], n)
                return [(float(x), float(y), float(x + k), float(y + k)) for (x, y), k in zip(xy, s)]
            tb, db = boxes(nt), boxes(nd)
            trackers = {10 + i: T(b) for i, b in enumerate(tb)}
            want = tbd._associate(trackers, [tuple(int(v) for v in b) for b in db])
            got = {d: 10 + t for t, d in _lib.associate(tb, db, 0.5)}
            assert got == want, (tb, db)


@needs_reference
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("params", [dict(every=0.0, ratio=0.3, gap=0.0, conf=10.), dict(every=0.0, ratio=0.5, gap=1.0, conf=10.),
                                    dict(every=0.12, ratio=0.5, gap=1.0, conf=10.)])
def test_reference_tracking_by_detection_equals_product(seed, params):
    """pyannote/video/tracking.py executed verbatim (API defaults, CLI defaults, detect_every > 0) == product state machine"""
    frames, dets = scenario(300 + seed, n=60, p_miss=0.3)
    det_of = {id(f): d for f, d in zip(frames, dets)}
    shots = [refhost._Segment(0, 0.8), refhost._Segment(0.8, 1.64), refhost._Segment(1.64, 2.4)]
    kw = dict(detect_every=params["every"], track_min_confidence=params["conf"], track_min_overlap_ratio=params["ratio"],
              track_max_gap=params["gap"])
    with refhost.reference_modules(_scripted_dlib()) as ref:
        want = list(ref.tracking.TrackingByDetection(lambda f: det_of[id(f)], **kw)(ScriptVideo(frames), shots))
    got = list(TrackingByDetection(lambda f: det_of[id(f)], trackers=ObjectTrackers(ScriptTracker), **kw)(ScriptVideo(frames), shots))
    assert got == want
    assert len(got) >= 2
