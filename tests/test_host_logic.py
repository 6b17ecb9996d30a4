"""CPU: the product's lock-step tracking state machine against the sequential restatement of the reference
(oracle/ref_flow.py), with scripted detections and a scripted tracker plugged into the reference's own seams
(detect_func: tracking.py:104,112,426; tracker object start_track/update/get_position: :203,231,250-251)."""
import os
import numpy as np
import pytest
from pyannote_video_amd.tracking_by_detection import TrackingByDetection, ObjectTrackers, get_segment_generator
from pyannote_video_amd import formats, pipeline
from pyannote_video_amd._core import Segment


class ScriptFrame(object):
    """stands for a frame: carries its index and the true boxes"""
    def __init__(self, i, boxes, shape=(360, 640, 3)):
        self.i, self.boxes, self.shape = i, boxes, shape


class ScriptTracker(object):
    """follows the nearest true box with a lag; confidence drops when nothing is near"""
    def __init__(self):
        self.box = None

    def start_track(self, frame, box):
        self.box = tuple(box.as_tuple() if hasattr(box, "as_tuple") else box)

    def update(self, frame):
        cx, cy = (self.box[0] + self.box[2]) / 2, (self.box[1] + self.box[3]) / 2
        best, bd = None, 1e9
        for b in frame.boxes:
            d = abs((b[0] + b[2]) / 2 - cx) + abs((b[1] + b[3]) / 2 - cy)
            if d < bd:
                best, bd = b, d
        if best is None or bd > 60:
            return 3.0
        self.box = tuple(0.5 * self.box[k] + 0.5 * best[k] + 0.25 * ((frame.i * 7 + k) % 3) for k in range(4))
        return 12.0 + (frame.i % 5) - 0.05 * bd

    def get_position(self):
        class R(object):
            def __init__(s, b): s.b = b
            def left(s): return s.b[0]
            def top(s): return s.b[1]
            def right(s): return s.b[2]
            def bottom(s): return s.b[3]
            def __iter__(s): return iter(s.b)
        return R(self.box)


class RefTracker(ScriptTracker):
    def get_position(self):
        return self.box


def scenario(seed, n=40, faces=3, p_miss=0.3, p_false=0.05):
    rng = np.random.default_rng(seed)
    frames, dets = [], []
    pos = rng.uniform(60, 300, (faces, 2)); size = rng.uniform(40, 90, faces)
    alive = [(int(rng.integers(0, n // 3)), int(rng.integers(2 * n // 3, n))) for _ in range(faces)]
    for i in range(n):
        pos += rng.normal(0, 2.0, pos.shape)
        boxes = [tuple(int(round(v)) for v in (pos[k, 0] - size[k] / 2, pos[k, 1] - size[k] / 2, pos[k, 0] + size[k] / 2, pos[k, 1] + size[k] / 2))
                 for k in range(faces) if alive[k][0] <= i <= alive[k][1]]
        d = [b for b in boxes if rng.random() > p_miss]
        if rng.random() < p_false:
            d.append((500, 20, 560, 80))
        frames.append(ScriptFrame(i, boxes)); dets.append(d)
    return frames, dets


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("params", [dict(ratio=0.3, gap=0.0), dict(ratio=0.5, gap=1.0)])
def test_shot_tracks_equal_sequential_reference(seed, params):
    from oracle import ref_flow
    frames, dets = scenario(seed)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    det_of = {id(f): d for f, d in zip(frames, dets)}
    tbd = TrackingByDetection(detect_func=lambda f: det_of[id(f)], track_min_overlap_ratio=params["ratio"], track_max_gap=params["gap"],
                              trackers=ObjectTrackers(ScriptTracker))
    got = tbd.process_shots([(cache, [True] * len(cache))], tbd._backend())[0]
    ref = ref_flow.track_shot(cache, dets, RefTracker, 10., params["ratio"], params["gap"])
    assert got == ref
    assert len(got) >= 1


def test_several_shots_in_lockstep_equal_one_by_one():
    shots, refs = [], []
    from oracle import ref_flow
    for seed in (11, 12, 13):
        frames, dets = scenario(seed, n=25 + seed)
        times = [100 * seed + i / 25.0 for i in range(len(frames))]
        cache = list(zip(times, frames))
        shots.append((cache, dets))
        refs.append(ref_flow.track_shot(cache, dets, RefTracker, 10., 0.5, 1.0))
    table = {id(f): d for cache, dets in shots for (_, f), d in zip(cache, dets)}
    tbd = TrackingByDetection(detect_func=lambda f: table[id(f)], track_min_overlap_ratio=0.5, track_max_gap=1.0,
                              trackers=ObjectTrackers(ScriptTracker))
    got = tbd.process_shots([(cache, [True] * len(cache)) for cache, _ in shots], tbd._backend())
    assert got == refs


class SpeculatingObjectTrackers(ObjectTrackers):
    """per-object trackers behind the bulk start / first-update path the GPU backend uses"""
    from pyannote_video_amd.tracking_by_detection import HipTrackers as _H
    speculate = _H.speculate


@pytest.mark.parametrize("seed", range(6))
def test_speculative_bulk_path_equals_sequential_reference(seed):
    from oracle import ref_flow
    frames, dets = scenario(100 + seed, n=50, p_miss=0.35)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0,
                              trackers=SpeculatingObjectTrackers(ScriptTracker))
    got = tbd.process_shots([(cache, [True] * len(cache), dets)], tbd._backend())[0]
    ref = ref_flow.track_shot(cache, dets, RefTracker, 10., 0.5, 1.0)
    assert got == ref


class ScriptVideo(object):
    def __init__(self, frames, fps=25.0, size=(640, 360)):
        self.frames, self.frame_rate, self.size, self.frame_size = frames, fps, size, size

    def __iter__(self):
        for i, f in enumerate(self.frames):
            yield i / self.frame_rate, f


@pytest.mark.parametrize("every", [0.0, 0.2])
def test_call_contract_shots_and_detect_every(every):
    from oracle import ref_flow
    frames, dets = scenario(21, n=60)
    det_of = {id(f): d for f, d in zip(frames, dets)}
    shots = [Segment(0, 0.8), Segment(0.8, 1.64), Segment(1.64, 2.4)]
    tbd = TrackingByDetection(detect_func=lambda f: det_of[id(f)], detect_every=every, track_min_overlap_ratio=0.5, track_max_gap=1.0,
                              trackers=ObjectTrackers(ScriptTracker))
    got = list(tbd(ScriptVideo(frames), shots))
    times = [i / 25.0 for i in range(len(frames))]
    ref = ref_flow.track_video(frames, times, [(s.start, s.end) for s in shots], lambda f: det_of[id(f)], RefTracker, 25.0,
                               detect_every=every, min_conf=10., ratio=0.5, max_gap=1.0)
    assert got == ref
    # a frame exactly at a segment end belongs to the next shot (tracking.py:44-58,406-417)
    assert pipeline.split_into_shots(times, [(s.start, s.end) for s in shots]) == [(0, 20), (20, 41), (41, 60)]


def test_extract_time_sync_equals_reference_generator():
    from oracle import ref_flow
    rng = np.random.default_rng(3)
    for fps in (25.0, 50.0, 23.976):
        times = [i / fps for i in range(40)]
        tracks = []
        for ident in range(4):
            a, b = sorted(rng.integers(0, 40, 2))
            tracks.append([(times[i], tuple(rng.uniform(0.1, 0.9, 4)), 'detection') for i in range(a, b + 1)])
        lines = ref_flow.track_text(tracks)
        frames = [np.zeros((100, 200, 3), np.uint8)] * 40
        seen = []
        ref_flow.extract(lines, frames, times, lambda f, box: seen.append(box) or np.zeros((68, 2), int), lambda f, p: np.zeros(128))
        rows = []
        for identifier, track in enumerate(tracks):
            for t, box, status in track:
                rows.append((formats.quantise_time(t), identifier, tuple(np.float32("%.3f" % v) for v in box), status))
        rows = [rows[i] for i in formats.pandas_sort_order([r[0] for r in rows])]       # tracking.sort_values('t')
        got = [box for _, _, g in pipeline.faces_per_frame(rows, times, 200, 100) for _, box in g]
        assert got == seen


def test_text_quantisation_helpers():
    rng = np.random.default_rng(4)
    e = rng.normal(0, 0.3, 4096).astype(np.float32)
    assert np.array_equal(np.round(e.astype(np.float64), 5), formats.quantise_embedding(e))
    box = (0.123456, 0.5, 0.987654, 0.75)
    line = list(formats.track_lines(3, [(1.23456, box, 'detection')]))[0]
    assert line == '1.235 3 0.123 0.500 0.988 0.750 detection\n'
    assert formats.quantise_track_box(box, 1920, 1080) == (int(float(np.float32('0.123')) * 1920), 540, int(float(np.float32('0.988')) * 1920), 810)
    # the product runs in float64 on the float32-parsed value, as pandas' iterrows() hands it to the reference (pyannote-face.py:142);
    # a float32 product would give 112 here
    assert formats.quantise_track_box((0.175, 0.175, 0.175, 0.175), 640, 640) == (111, 111, 111, 111)


class _FakeExtractCtx(object):
    def landmarks(self, frames, boxes):
        return np.array([[[b[0] + f, b[1] + f]] * 68 for f, b in zip(frames, boxes)], np.int32).reshape(-1, 68, 2)

    def embed(self, frames, pts):
        return np.array([[float(f) + p[0, 0] * 1e-3] * 128 for f, p in zip(frames, pts)], np.float32).reshape(-1, 128)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("drop_last", [True, False])
def test_extract_stream_equals_whole_file_walk(seed, drop_last):
    """shot-by-shot extraction == the reference's walk over the finished track file (including its lagging frame pointer and
    the group it never yields)"""
    from pyannote_video_amd import pipeline, formats
    rng = np.random.default_rng(seed)
    fps = 25.0
    n_frames = 90
    # frame times that do not always survive the 3-decimal quantisation (so that groups can lag behind frames)
    times = [i / fps + (1e-4 if i % 7 == 3 else 0.0) for i in range(n_frames)]
    shots = [(0, 30), (30, 52), (52, 90)]
    per_shot = []
    for a, b in shots:
        tracks = []
        for _ in range(int(rng.integers(0, 4))):
            i0 = int(rng.integers(a, b - 2)); i1 = int(rng.integers(i0 + 1, b))
            box = rng.uniform(0.1, 0.6, 2)
            tracks.append([(times[i], (box[0], box[1], box[0] + 0.2, box[1] + 0.3), "detection") for i in range(i0, i1)])
        per_shot.append(tracks)
    W, H = 640, 360
    frames = list(range(n_frames))
    ex = pipeline.ExtractStream(_FakeExtractCtx(), frames, times, W, H)
    for tracks in per_shot:
        ex.feed(tracks)
    pts, emb = ex.finish(drop_last=drop_last)
    # whole-file form
    rows = []
    for identifier, track in enumerate(t for tracks in per_shot for t in tracks):
        for t, box, status in track:
            rows.append((formats.quantise_time(t), identifier, tuple(np.float32("%.3f" % v) for v in box), status))
    rows = [rows[i] for i in formats.pandas_sort_order([r[0] for r in rows])]           # tracking.sort_values('t') of the whole file
    ref = pipeline.faces_per_frame(rows, times, W, H, drop_last=drop_last)
    assert ex.emitted == [(fi, T) for fi, T, _ in ref]
    assert ex.face_id == [ident for _, _, g in ref for ident, _ in g]
    assert ex.face_boxes == [box for _, _, g in ref for _, box in g]
    assert ex.rows == rows
    assert len(emb) == len(ex.face_id) == len(pts)


class ModelScriptTracker(ScriptTracker):
    """a scripted tracker with a 'model' (age) that every full update advances and that feeds back into the confidence, so a
    forgotten or misplaced commit of a deferred update changes the tracks"""
    def __init__(self):
        ScriptTracker.__init__(self)
        self.age = 0

    def update(self, frame):
        conf = ScriptTracker.update(self, frame)
        conf = conf - 0.7 * (self.age % 4)
        self.age += 1
        return conf


class ModelRefTracker(ModelScriptTracker):
    def get_position(self):
        return self.box


class DeferringObjectTrackers(SpeculatingObjectTrackers):
    """per-object trackers behind the deferred-update interface of the GPU backend (update_many(defer=True) / commit_many)"""
    def __init__(self, factory):
        SpeculatingObjectTrackers.__init__(self, factory)
        self.pending = {}
        self.commits = 0

    def update_many(self, handles, frames, defer=False):
        import copy
        if not defer:
            assert not any(id(h) in self.pending for h in handles), "update of a tracker with an uncommitted deferred update"
            return ObjectTrackers.update_many(self, handles, frames)
        psr, pos = [], []
        for h, f in zip(handles, frames):
            probe = copy.deepcopy(h)
            psr.append(probe.update(f))
            p = probe.get_position()
            pos.append((p.left(), p.top(), p.right(), p.bottom()))
            self.pending[id(h)] = (f, h)
        return np.array(psr, np.float64), np.array(pos, np.float64).reshape(-1, 4)

    def commit_many(self, handles, frames):
        for h, f in zip(handles, frames):
            g, _ = self.pending.pop(id(h))
            assert g is f, "commit with a different frame than the deferred update ran on"
            h.update(f)
            self.commits += 1

    def release(self, handle):
        self.pending.pop(id(handle), None)


@pytest.mark.parametrize("seed", range(6))
def test_deferred_first_updates_with_commit_equal_sequential_reference(seed):
    from oracle import ref_flow
    frames, dets = scenario(400 + seed, n=60, p_miss=0.4)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    backend = DeferringObjectTrackers(ModelScriptTracker)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=backend)
    got = tbd.process_shots([(cache, [True] * len(cache), dets)], backend)[0]
    ref = ref_flow.track_shot(cache, dets, ModelRefTracker, 10., 0.5, 1.0)
    assert got == ref
    assert backend.commits > 0          # the scenario has survivors, i.e. the commit path ran


def test_inline_overlap_gate_equals_rectangle_form():
    """_no_match (used by _fix / _fill_gaps) == (_match(drectangle, drectangle) == 0) on random, touching, empty and nested boxes"""
    from pyannote_video_amd.shim import drectangle
    rng = np.random.default_rng(9)
    for ratio in (0.0, 0.3, 0.5, 1.0):
        tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=ratio)
        for _ in range(3000):
            a = rng.integers(-5, 40, 4).astype(float) * rng.choice([1.0, 0.5])
            b = rng.integers(-5, 40, 4).astype(float) * rng.choice([1.0, 0.5])
            if rng.random() < 0.3:
                b = a + rng.integers(-2, 3, 4)
            if rng.random() < 0.7:
                a[2:] = np.maximum(a[2:], a[:2]); b[2:] = np.maximum(b[2:], b[:2])
            want = tbd._match(drectangle(*a), drectangle(*b)) == 0
            assert tbd._no_match(tuple(a), tuple(b)) == want, (ratio, a, b)


def test_frames_staged_for_one_call_are_not_evicted_before_it_runs():
    """one batched call may reference more distinct numpy frames than the staging cache holds (a 4096-tracker batch of
    a long shot): nothing is released until the call has run, then the cache shrinks back to its capacity"""
    from pyannote_video_amd.runtime import Context

    class FakeFrame(object):
        def __init__(self, k): self.handle, self.keep, self.released = k, None, False
        def release(self): self.released = True

    c = object.__new__(Context)
    c._h, c._staged, c._staged_order, c.stage_capacity, c._hold = None, {}, [], 4, 0
    made = []

    def upload(rgb):
        made.append(FakeFrame(len(made) + 1))
        return made[-1]
    c.upload = upload
    frames = [np.zeros((2, 2, 3), np.uint8) for _ in range(10)]
    with c._staging():
        hs = c._handles(frames)
        assert list(hs) == list(range(1, 11))
        assert not any(f.released for f in made)          # all ten are still resident while the C call would run
    assert [f.released for f in made] == [True] * 6 + [False] * 4
    assert len(c._staged) == 4
    c._handles(frames[6:])                                # cached ones are reused, not uploaded again
    assert len(made) == 10


def test_round3_equals_text_round_trip():
    """formats.quantise_time / ExtractStream.prepare use round(v, 3) for the '%.3f' round trip of the track file"""
    rng = np.random.default_rng(9)
    vals = np.concatenate([rng.random(200000), rng.random(50000) * 40, (np.arange(100000) + 0.5) / 1000.0, (np.arange(100000) + 0.5) / 1000.0 + 1e-17])
    for v in vals.tolist():
        assert round(v, 3) == float("%.3f" % v)
    box = (0.123456, 0.5, 0.987654, 0.75)
    q = np.asarray([round(v, 3) for v in box], np.float64).astype(np.float32).astype(np.float64).tolist()
    assert q == [float(np.float32("%.3f" % v)) for v in box]
    # numpy scalars (times = np.arange(n) / fps): round() of an np.float64 is numpy's scale / rint / divide, not the correctly rounded
    # value -- np.float64(0.1125) gives 0.112 where '%.3f' gives 0.113 -- so the product converts to a Python float first
    assert round(np.float64(0.1125), 3) != float("%.3f" % 0.1125)
    for fps in (29.97, 23.976, 25.0, 30.0):
        for t in np.arange(3000) / fps:
            assert formats.quantise_time(t) == float("%.3f" % t)
    ex = pipeline.ExtractStream(None, [None] * 4, [0.0, 0.1125, 0.2, 0.3], 100, 100)
    ex.prepare([[(np.float64(0.1125), tuple(np.float64(v) for v in (0.1125, 0.2, 0.3125, 0.4)), "detection"), (np.float64(0.2), (0.1, 0.2, 0.3, 0.4), "detection")]])
    assert ex.file_T == [0.113, 0.2] and ex.rows_file[0][2][0] == float(np.float32("0.113"))


def test_text_rows_formatted_by_the_library_equal_python_formatting():
    """formats.landmark_rows / embedding_rows (pvf_format_rows) == the reference's per-value '{:.5f}' formatting (pyannote-face.py:299-311),
    byte for byte, rounding ties and negative zeros included"""
    rng = np.random.default_rng(4)
    emb = rng.normal(0, 0.1, (300, 128)).astype(np.float32)
    emb[0, :8] = [0.000005, -0.000005, 0.123455, 1e-9, -1e-9, 0.0, -0.0, 123456.7]
    emb[1, :4] = [0.5e-5, 1.5e-5, 2.5e-5, -2.5e-5]
    T = np.round(rng.random(300) * 40, 3)
    ids = rng.integers(0, 500, 300)
    want = "".join(formats.embedding_line(float(T[i]), int(ids[i]), emb[i]) for i in range(300)).encode()
    assert formats.embedding_rows(T, ids, emb) == want
    pts = rng.integers(0, 1920, (300, 68, 2)).astype(np.int32)
    want = "".join(formats.landmark_line(float(T[i]), int(ids[i]), pts[i], 1920, 1080) for i in range(300)).encode()
    assert formats.landmark_rows(T, ids, pts, 1920, 1080) == want
    from pyannote_video_amd import _lib
    x = np.concatenate([(rng.integers(-10**7, 10**7, 20000) + 0.5) / 1e5, rng.normal(0, 1, 20000), rng.normal(0, 1e3, 10000), [1e12, -3e15]])
    assert _lib.format_rows(np.zeros(1), np.zeros(1, np.int64), x.reshape(1, -1)) == ("0.000 0" + "".join(" %.5f" % v for v in x.tolist()) + "\n").encode()
    assert formats.embedding_rows([], [], np.zeros((0, 128))) == b""
    # values that would not fit a row's fixed slot (27+ digits before the point) are refused before anything is written (they
    # used to run past the row slot and, on the last row, past the buffer); NaN / inf print like Python's format
    for big in (1e25, -3e38, 1e300):
        with pytest.raises(Exception, match="does not fit"):
            _lib.format_rows(np.zeros(2), np.zeros(2, np.int64), np.array([[0.5, 1.0], [big, 2.0]]))
    with pytest.raises(Exception, match="does not fit"):
        _lib.format_rows(np.array([1e30]), np.zeros(1, np.int64), np.zeros((1, 2)))
    assert _lib.format_rows(np.zeros(1), np.zeros(1, np.int64), np.array([[np.nan, np.inf, -np.inf, 9.9e24]])) == \
        ("0.000 0" + "".join(" %.5f" % v for v in [float("nan"), float("inf"), float("-inf"), 9.9e24]) + "\n").encode()


@pytest.mark.parametrize("seed", range(6))
def test_tracks_without_the_graph_equal_tracks_from_the_graph(seed):
    """finish_shot (union-find over the recorded edges, in the graph's node order) == finish_shot_graph (networkx, the reference's
    own structure): same tracks, same order"""
    frames, dets = scenario(100 + seed, n=60, faces=4, p_miss=0.4, p_false=0.15)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=ObjectTrackers(ScriptTracker))
    backend = tbd._backend()
    job = tbd.begin_shot(cache, [True] * len(cache), dets, backend)
    tbd._run_lanes(job["lanes"], backend)
    fast = tbd.finish_shot(job)
    slow = tbd.finish_shot_graph(job)
    assert fast == slow and len(fast) >= 2


class FakeTrackerContext(object):
    """the slice of runtime.Context that HipTrackers uses, on scripted CPU trackers: handles are integers, frames are looked up through
    uint64 handle arrays, clones are deep copies, deferred updates leave the tracker's model untouched until commit"""
    def __init__(self, factory):
        self.factory, self.frames, self.trk, self.pending, self.next = factory, {}, {}, {}, 1
        self.clones = self.commits = 0

    def frame_handles(self, frames):
        out = np.zeros(len(frames), np.uint64)
        for i, f in enumerate(frames):
            self.frames[1000 + id(f) % 10**9] = f
            out[i] = 1000 + id(f) % 10**9
        return out

    def _frames(self, fh):
        return [self.frames[int(h)] for h in fh] if isinstance(fh, np.ndarray) else list(fh)

    def tracker_create_many(self, n, as_array=False):
        ids = np.arange(self.next, self.next + n, dtype=np.uint64)
        self.next += n
        for i in ids.tolist():
            self.trk[i] = None
        return ids if as_array else ids.tolist()

    def tracker_start_many(self, trks, frames, boxes):
        for h, f, b in zip(np.asarray(trks).tolist(), self._frames(frames), np.asarray(boxes, np.float64).reshape(-1, 4).tolist()):
            t = self.factory()
            t.start_track(f, tuple(b))
            self.trk[h] = t

    def tracker_clone_many(self, trks, as_array=False):
        import copy
        ids = self.tracker_create_many(len(trks), as_array=True)
        for s, d in zip(np.asarray(trks).tolist(), ids.tolist()):
            self.trk[d] = copy.deepcopy(self.trk[s])
            self.clones += 1
        return ids if as_array else ids.tolist()

    def tracker_update_many(self, trks, frames, defer=False):
        import copy
        psr, pos = [], []
        for h, f in zip(np.asarray(trks).tolist(), self._frames(frames)):
            assert h not in self.pending, "update of a tracker with an uncommitted deferred update"
            t = copy.deepcopy(self.trk[h]) if defer else self.trk[h]
            psr.append(t.update(f))
            p = t.get_position()
            pos.append((p.left(), p.top(), p.right(), p.bottom()))
            if defer:
                self.pending[h] = f
        return np.array(psr, np.float64), np.array(pos, np.float64).reshape(-1, 4)

    def tracker_commit_many(self, trks, frames):
        for h, f in zip(np.asarray(trks).tolist(), self._frames(frames)):
            assert self.pending.pop(h) is f, "commit with a different frame than the deferred update ran on"
            self.trk[h].update(f)
            self.commits += 1

    def tracker_destroy(self, h):
        self.pending.pop(int(h), None)
        del self.trk[int(h)]

    def tracker_destroy_many(self, hs):
        for h in hs:
            self.tracker_destroy(h)


@pytest.mark.parametrize("seed", range(6))
def test_bulk_pair_path_on_handle_arrays_equals_sequential_reference(seed):
    """HipTrackers.speculate_pair (starts once, clones for the backward pass, deferred first updates addressed through handle / owner /
    box arrays) feeding the two lanes of a shot == the frame-by-frame reference flow"""
    from oracle import ref_flow
    from pyannote_video_amd.tracking_by_detection import HipTrackers
    frames, dets = scenario(700 + seed, n=60, faces=4, p_miss=0.4, p_false=0.1)
    times = [i / 25.0 for i in range(len(frames))]
    cache = list(zip(times, frames))
    ctx = FakeTrackerContext(ModelScriptTracker)
    backend = HipTrackers(ctx)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=backend)
    det_at = {t: d for (t, _), d in zip(cache, dets)}
    plans = backend.speculate_pair(cache, det_at)
    job = tbd.begin_shot(cache, [True] * len(cache), dets, backend, plans)
    tbd._run_lanes(job["lanes"], backend)
    got = tbd.finish_shot(job)
    ref = ref_flow.track_shot(cache, dets, ModelRefTracker, 10., 0.5, 1.0)
    assert got == ref
    assert ctx.clones == sum(len(d) for d in dets) and ctx.commits > 0
    assert not ctx.trk                      # every tracker was released


def test_hac_quotient_by_reciprocal_is_the_ieee_quotient(tmp_path):
    """hac_persist_k divides by the (merge-uniform) sum of two sizes through one reciprocal and two fused corrections: 9.6e7 cases against
    the division itself (tests/csrc/exact_quotient_check.c)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "exact_quotient_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "exact_quotient_check.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, src, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-400:]
    assert " 0 differ" in out.stdout


def test_parse_rows_returns_what_loadtxt_returns():
    """pvf_parse_rows (the embedding.txt reader behind FaceClustering's preprocess) against np.loadtxt: the writers' 5-decimal rows bit for
    bit, exponents / inf / nan / long digit strings through strtod, blank lines skipped, ragged rows and non-numbers refused"""
    import io
    from pyannote_video_amd import _lib
    rng = np.random.default_rng(3)
    n = 300
    t = np.round(rng.uniform(0, 4000, n), 3)
    ident = rng.integers(0, 5000, n)
    v = rng.normal(size=(n, 128)) * rng.choice([1e-6, 1e-2, 1.0, 1e3], size=(n, 1))
    v[0, :4] = [0.0, -0.0, -1e-7, 12345.678901]
    for decimals in (5, 3, 9):
        txt = _lib.format_rows(t, ident, v, decimals)
        a = _lib.parse_rows(txt)
        b = np.loadtxt(io.BytesIO(txt), dtype=np.float64, ndmin=2)
        assert a.shape == b.shape == (n, 130)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    txt = b"1e-3 2.5E+2 -inf nan +3\n\n0.1234567890123456789 7 .5 9. 123456789012345678\r\n"
    a, b = _lib.parse_rows(txt), np.loadtxt(io.BytesIO(txt), ndmin=2)
    assert a.shape == (2, 5) and np.array_equal(a, b, equal_nan=True)
    assert _lib.parse_rows(b"").shape == (0, 0)
    assert _lib.parse_rows(b"1 2 3 4 5 6 7 8 9\n").shape == (1, 9)          # (more values than the first buffer guess holds)
    for bad in (b"1 2\n3\n", b"1 x\n", b"1 2-3\n"):
        with pytest.raises(_lib.PvfError):
            _lib.parse_rows(bad)


def test_read_embeddings_round_trip(tmp_path):
    from pyannote_video_amd import _lib
    rng = np.random.default_rng(4)
    t = np.round(rng.uniform(0, 40, 50), 3)
    ident = rng.integers(0, 9, 50)
    v = rng.normal(size=(50, 128)) * 0.1
    p = tmp_path / "embedding.txt"
    p.write_bytes(_lib.format_rows(t, ident, v, 5))
    tt, ii, xx = formats.read_embeddings(str(p))
    assert np.array_equal(tt, t) and np.array_equal(ii, ident) and xx.flags.c_contiguous
    assert np.array_equal(xx, np.round(v, 5))
    (tmp_path / "empty.txt").write_bytes(b"")
    tt, ii, xx = formats.read_embeddings(str(tmp_path / "empty.txt"))
    assert len(tt) == 0 and xx.shape == (0, 128)


def test_in_memory_clustering_plan_equals_preprocess_and_call():
    """FaceClustering.cluster_rows (float32 rows in memory, table made on the device) hands the kernels exactly the rows, in exactly the
    order, that preprocess() + __call__ hand them for the float64 table (clustering.py:59-83: (track, time) order, single-timestamp
    tracks left out), with and without a source index; the labels are mapped back the same way"""
    from pyannote_video_amd.clustering import FaceClustering
    rng = np.random.default_rng(8)

    class Rec(object):
        def cluster_tracks(self, X, row_start, cut):
            self.f64 = (np.array(X), np.array(row_start), cut)
            T = len(row_start) - 1
            return np.arange(T) // 2 * 2, np.array([[2 * k, 2 * k + 1, 0.25, 2.0] for k in range(T // 2)])

        def cluster_tracks_f32(self, emb, order, row_start, cut, decimals=5, metric=0):
            self.f32 = (np.round(np.asarray(emb)[order].astype(np.float64), decimals), np.array(row_start), cut, metric)
            T = len(row_start) - 1
            return np.arange(T) // 2 * 2, np.array([[2 * k, 2 * k + 1, 0.25, 2.0] for k in range(T // 2)])

    for trial in range(20):
        n_tracks = int(rng.integers(1, 12))
        time, track = [], []
        for t in range(n_tracks):
            n = int(rng.integers(1, 6))
            t0 = float(rng.integers(0, 100)) / 25.0
            single = rng.random() < 0.25
            for k in range(1 if single else n):
                time.append(t0 + k / 25.0); track.append(3 * t + 1)        # (ids with gaps)
        perm = rng.permutation(len(time))
        time, track = np.array(time)[perm], np.array(track)[perm]
        emb = rng.normal(0, 0.1, (len(time), 128)).astype(np.float32)
        rec = Rec()
        fc = FaceClustering(ctx=rec)
        sp, feats = fc.model.preprocess((time, track, np.round(emb.astype(np.float64), 5)))
        want = fc(sp, features=feats)
        want_labels = {int(t): int(l) for _, t, l in want.itertracks(yield_label=True)}
        hist = fc.history
        got = fc.cluster_rows(time, track, emb)
        if not want_labels:
            assert got == {} and not hasattr(rec, "f32")
            continue
        assert got == want_labels and fc.history == hist
        assert np.array_equal(rec.f32[0], rec.f64[0]) and np.array_equal(rec.f32[1], rec.f64[1]) and rec.f32[2] == rec.f64[2]
        # rows that were reordered without being moved (a gathered table): row k of (time, track) is emb2[src[k]]
        src = rng.permutation(len(time))
        emb2 = np.zeros_like(emb); emb2[src] = emb
        assert fc.cluster_rows(time, track, emb2, src_index=src) == want_labels and np.array_equal(rec.f32[0], rec.f64[0])
