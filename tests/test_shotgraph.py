"""CPU: the library's host state machine of a shot (csrc/shotgraph.hip: pvf_lane_*, pvf_shot_tracks, pvf_track_rows, pvf_round_decimals)
against the Python form it replaces on the engine's critical path (tracking_by_detection._lane / finish_shot, engine.ExtractStream.prepare)
-- which tests/refhost.py and tests/test_host_logic.py hold against the reference's tracking.py executed verbatim and against the oracle
flow.  Scripted trackers on CPU; every add_edge call of both passes, every track row and every number of the track file must be equal.
Reference: pyannote/video/tracking.py:184-357, scripts/pyannote-face.py:125-127,142-145,262-266."""
import itertools

import numpy as np
import pytest

from pyannote_video_amd import _lib, engine, formats
from pyannote_video_amd.tracking_by_detection import TrackingByDetection, HipTrackers, NativeLane, status_of, detection_arrays
from test_host_logic import scenario, FakeTrackerContext, ModelScriptTracker, ModelRefTracker


def _both(seed, n, faces, p_miss, p_false, ratio, gap, dup=False, every=1):
    frames, dets = scenario(seed, n=n, faces=faces, p_miss=p_miss, p_false=p_false)
    if every > 1:
        dets = [d if i % every == 0 else [] for i, d in enumerate(dets)]
    if dup:                                                   # a box seen twice on a frame is ONE node of the reference's graph
        dets = [d + d[:1] if (i % 7 == 3 and d) else d for i, d in enumerate(dets)]
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    out = []
    for python_lanes in (False, True):
        ctx = FakeTrackerContext(ModelScriptTracker)
        backend = HipTrackers(ctx)
        tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=ratio, track_max_gap=gap, trackers=backend)
        tbd.python_lanes = python_lanes
        det_at = {t: d for (t, _), d in zip(cache, dets)}
        plans = backend.speculate_pair(cache, det_at)
        job = tbd.begin_shot(cache, [True] * len(cache), dets, backend, plans)
        assert ("native" in job) == (not python_lanes)
        tbd._run_lanes(job["lanes"], backend)
        if not python_lanes:
            view = dict(job)
            tbd._python_view(view)
            edges = (view["ef"], view["eb"])
        else:
            edges = (job["ef"], job["eb"])
        tracks = tbd.finish_shot(job)
        assert not ctx.trk                                    # every tracker was released
        out.append((edges, tracks, job, ctx.commits))
    return cache, dets, out


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("params", [dict(ratio=0.3, gap=0.0, p_miss=0.4, p_false=0.1), dict(ratio=0.5, gap=1.0, p_miss=0.25, p_false=0.2),
                                    dict(ratio=0.5, gap=0.2, p_miss=0.0, p_false=0.0), dict(ratio=0.3, gap=1.0, p_miss=0.6, p_false=0.3, dup=True),
                                    dict(ratio=0.5, gap=1.0, p_miss=0.2, p_false=0.1, every=5)])
def test_native_passes_and_tracks_equal_the_python_form(seed, params):
    p = dict(params)
    cache, dets, (nat, py) = _both(900 + seed, n=50 + 3 * seed, faces=2 + seed % 4, p_miss=p.pop("p_miss"), p_false=p.pop("p_false"), **p)
    # every add_edge call of both passes: nodes (t, box, status) and confidences, in order
    assert nat[0][0] == py[0][0] and nat[0][1] == py[0][1]
    assert nat[1] == py[1] and len(nat[1]) >= 1                # the tracks: rows (t, integer box, status string), track order
    assert nat[3] == py[3]                                    # the same deferred updates were committed
    if params["p_miss"] > 0:
        assert any(st != "detection" for tr in nat[1] for _, _, st in tr)      # (trackers did bridge missed detections)


def test_native_tracks_equal_the_oracle_flow():
    from oracle import ref_flow
    for seed in range(6):
        cache, dets, (nat, py) = _both(40 + seed, n=60, faces=4, p_miss=0.4, p_false=0.1, ratio=0.5, gap=1.0)
        ref = ref_flow.track_shot(cache, dets, ModelRefTracker, 10., 0.5, 1.0)
        assert nat[1] == ref


def test_status_codes():
    assert status_of(1 << 8) == "detection"
    assert status_of(1 | (1 << 8) | (1 << 16)) == "forward+detection+backward"
    assert status_of(2 | (1 << 16) | (1 << 24)) == "error(forward+forward+backward)"


def test_round_decimals_is_pythons_round():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 4000, 20000), rng.uniform(0, 1, 20000), np.arange(0, 2000) / 1920.0, np.arange(0, 1100) / 1080.0,
                        np.arange(100000) / 25.0 % 977.0, (np.arange(4000) + 0.5) / 1000.0, (np.arange(4000) + 0.5) / 1000.0 + 1e-12,
                        np.nextafter((np.arange(4000) + 0.5) / 1000.0, 0), [0.0, 0.1125, 2.675, 1e-9, 0.0005, 0.0015, 1234.5675]])
    x = np.concatenate([x, -x[:5000]])
    for k in (3, 5, 0):
        got = _lib.round_decimals(x, k)
        want = np.array([round(float(v), k) for v in x])
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), k


def test_track_rows_are_the_track_files_numbers():
    rng = np.random.default_rng(6)
    for (tw, th), (w, h) in (((1920, 1080), (1920, 1080)), ((640, 360), (1920, 1080)), ((1279, 719), (1279, 719)), ((3840, 2160), (3840, 2160))):
        b = np.stack([rng.integers(-40, tw + 40, 5000), rng.integers(-40, th + 40, 5000), rng.integers(-40, tw + 40, 5000), rng.integers(-40, th + 40, 5000)], 1)
        fb, pb = _lib.track_rows(b, tw, th, w, h)
        for row, f, p in zip(b.tolist(), fb.tolist(), pb.tolist()):
            norm = (row[0] / tw, row[1] / th, row[2] / tw, row[3] / th)
            q = [float(np.float32("%.3f" % v)) for v in norm]
            assert f == q
            assert tuple(p) == formats.denormalise(q, w, h)


@pytest.mark.parametrize("seed", range(4))
def test_prepare_rows_equals_prepare(seed):
    """ExtractStream.prepare_rows (arrays from the library) leaves the stream in the state ExtractStream.prepare (Python rows) does"""
    cache, dets, (nat, py) = _both(300 + seed, n=40, faces=3, p_miss=0.3, p_false=0.1, ratio=0.5, gap=1.0)
    job = nat[2]
    tw, th, w, h = 640, 360, 1280, 720
    times = [t for t, _ in cache]
    frames = [f for _, f in cache]
    norm = [TrackingByDetection._normalize_track(tr, tw, th) for tr in nat[1]]

    class Ctx(object):
        pass
    a = engine.ExtractStream(Ctx(), frames, times, w, h)
    b = engine.ExtractStream(Ctx(), frames, times, w, h)
    wa = a.prepare(norm)
    rows, starts = job["rows"], job["track_start"]
    nb = (rows[:, 1:5].astype(np.float64) / np.array([tw, th, tw, th], np.float64)).tolist()
    norm_b, i = [], 0
    for tr in nat[1]:
        norm_b.append([(t, tuple(nb[i + j]), st) for j, (t, _, st) in enumerate(tr)])
        i += len(tr)
    assert norm_b == norm
    wb = b.prepare_rows(norm_b, times, rows, starts, (tw, th))
    assert (wa[1], wa[2]) == (wb[1], wb[2]) and [id(f) for f in wa[0]] == [id(f) for f in wb[0]]
    assert a.rows_file == b.rows_file and a.file_T == b.file_T and a.file_id == b.file_id
    assert a.groups == b.groups and a.face_T == b.face_T and a.face_id == b.face_id and a.face_boxes == b.face_boxes
    assert all(type(x) is type(y) for ra, rb in zip(a.rows_file, b.rows_file) for x, y in zip(ra, rb))
    assert all(type(v) is int for _, g in b.groups for _, box in g for v in box)


def test_lane_requests_only_what_outlives_its_first_update():
    """with a detection for every face on every frame no tracker outlives its first update: a native pass ends without a single request"""
    frames, dets = scenario(7, n=30, faces=3, p_miss=0.0, p_false=0.0)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    ctx = FakeTrackerContext(ModelScriptTracker)
    backend = HipTrackers(ctx)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=backend)
    plans = backend.speculate_pair(cache, {t: d for (t, _), d in zip(cache, dets)})
    job = tbd.begin_shot(cache, [True] * len(cache), dets, backend, plans)
    for lane in job["lanes"]:
        assert list(lane) == []                               # the coroutine yields nothing
    tracks = tbd.finish_shot(job)
    assert all(st == "detection" for tr in tracks for _, _, st in tr)
    assert ctx.commits == 0 and not ctx.trk
