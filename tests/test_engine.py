"""CPU: the streaming engine (pyannote_video_amd/engine.py) on a scripted context -- scripted detections, the scripted model trackers of
tests/test_host_logic.py, deterministic landmark / embedding stand-ins.  What is checked is the engine's own work: the same tracks and
faces whichever way the shots flow (resident list, streamed numpy frames through the ingest ring, sequential or pipelined, one job or
many), windowed bulk tracker starts equal to whole-shot ones, and frames released shot by shot (reference tracking.py:359-362,410-420:
memory is bounded per shot)."""
import numpy as np
import pytest

from pyannote_video_amd import engine, pipeline
from pyannote_video_amd.tracking_by_detection import TrackingByDetection, HipTrackers
from tests.test_host_logic import FakeTrackerContext, ModelScriptTracker, ModelRefTracker, scenario


class FakeDeviceFrame(object):
    """what the fake ingest ring makes of a numpy frame: knows its scenario frame, counts its release"""
    live = 0
    peak = 0

    def __init__(self, script):
        self.script, self.released = script, False
        self.i, self.boxes, self.shape = script.i, script.boxes, script.shape
        FakeDeviceFrame.live += 1
        FakeDeviceFrame.peak = max(FakeDeviceFrame.peak, FakeDeviceFrame.live)

    def release(self):
        assert not self.released, "frame released twice"
        self.released = True
        FakeDeviceFrame.live -= 1


class FakeRing(object):
    def __init__(self, ctx, h, w):
        self.ctx, self.h, self.w = ctx, h, w

    def push(self, rgb):
        return FakeDeviceFrame(self.ctx.script_of[int(rgb[0, 0, 0]) * 256 + int(rgb[0, 0, 1])])

    def close(self):
        pass


class FakeContext(FakeTrackerContext):
    """FakeTrackerContext + the detector / landmark / embedding calls of runtime.Context the engine uses"""

    def __init__(self, frames, dets):
        FakeTrackerContext.__init__(self, ModelScriptTracker)
        self.update_calls = 0
        self.script_of = {f.i: f for f in frames}
        self.dets_of = {f.i: d for f, d in zip(frames, dets)}
        self.detect_calls = 0

    def ingest_ring(self, h, w, depth=8):
        return FakeRing(self, h, w)

    def tracker_update_many(self, trks, frames, defer=False):
        self.update_calls += 1
        return FakeTrackerContext.tracker_update_many(self, trks, frames, defer)

    def detect_many(self, frames, batch, upsample=1, adjust_threshold=0.0, cap=64, arrays=False):
        self.detect_calls += 1
        dets = [self.dets_of[f.i] for f in frames]
        m = max([len(d) for d in dets] + [0])
        out = np.zeros((len(frames), m, 4), np.int32)
        cnt = np.zeros(len(frames), np.int32)
        for k, d in enumerate(dets):
            cnt[k] = len(d)
            for j, b in enumerate(d):
                out[k, j] = b
        assert arrays
        return out, np.zeros((len(frames), m), np.float32), cnt

    def landmarks(self, frames, boxes):
        pts = np.zeros((len(boxes), 68, 2), np.int32)
        for k, (f, b) in enumerate(zip(frames, boxes)):
            assert not getattr(f, "released", False), "landmarks on a released frame"
            pts[k, :, 0] = b[0] + np.arange(68) + f.i
            pts[k, :, 1] = b[1] + 2 * np.arange(68)
        return pts

    def embed(self, frames, pts):
        out = np.zeros((len(pts), 128), np.float32)
        for k, (f, p) in enumerate(zip(frames, pts)):
            assert not getattr(f, "released", False), "embedding on a released frame"
            out[k] = np.sin(0.01 * (p[0, 0] + p[0, 1]) + 0.1 * np.arange(128)) * 0.1
        return out


def numpy_frame(f):
    a = np.zeros(f.shape, np.uint8)
    a[0, 0, 0], a[0, 0, 1] = f.i // 256, f.i % 256
    return a


def make_video(seed, n_shots=4, n=30, **kw):
    frames, dets, shots, t0 = [], [], [], 0
    for s in range(n_shots):
        fr, de = scenario(1000 * seed + s, n=n + 3 * s, **kw)
        for f in fr:
            f.i += t0
        frames += fr; dets += de
        shots.append((t0 / 25.0, (t0 + len(fr)) / 25.0))
        t0 += len(fr)
    times = [i / 25.0 for i in range(len(frames))]
    return frames, dets, times, shots


def run_engine(frames, dets, times, shots, mode, overlap=True, limit=8192, window=4096, extract=True, group=1, batch=7):
    ctx = FakeContext(frames, dets)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(ctx))
    eng = engine.Engine(ctx, tbd, detect_batch_size=batch, overlap=overlap, speculate_limit=limit, speculate_window=window, group=group)
    if mode == "resident":
        job = engine.VideoJob(ctx, 640, 360, frames=frames, times=times, extract=extract)
        src = engine.resident_source(job, frames, times, shots, 1)
        eng.run(src, HipTrackers(ctx), n_shots=len(engine.split_into_shots(times, shots)))
    else:
        job = engine.VideoJob(ctx, 640, 360, extract=extract)
        video = [(t, numpy_frame(f)) for t, f in zip(times, frames)]
        src = engine.StreamSource(ctx, [(job, video, shots, 1, None)])
        try:
            eng.run(src, HipTrackers(ctx))
        finally:
            src.close()
    assert not ctx.trk, "every tracker was released"
    if not extract:
        return job.tracks, None, None, ctx, eng
    pts, emb = job.ex.finish(computed=True)
    return job.ex.tracks, (job.ex.face_T, job.ex.face_id, job.ex.face_boxes), (pts, emb), ctx, eng


@pytest.mark.parametrize("seed", range(3))
def test_streamed_pipelined_equals_resident_sequential_and_the_reference_flow(seed):
    from oracle import ref_flow
    frames, dets, times, shots = make_video(seed, faces=4, p_miss=0.35, p_false=0.1)
    base = run_engine(frames, dets, times, shots, "resident", overlap=False)
    # the reference flow, shot by shot
    ref = []
    for i0, i1 in engine.split_into_shots(times, shots):
        cache = list(zip(times[i0:i1], frames[i0:i1]))
        for tr in ref_flow.track_shot(cache, dets[i0:i1], ModelRefTracker, 10., 0.5, 1.0):
            ref.append(TrackingByDetection._normalize_track(tr, 640, 360))
    assert base[0] == ref and len(ref) >= 8
    # `extract`'s pairing of faces and frames: the whole-file walk of the reference's generator
    rows = [(round(t, 3), k, tuple(np.float32("%.3f" % v) for v in box), st) for k, tr in enumerate(ref) for t, box, st in tr]
    rows.sort(key=lambda r: r[0])
    want = [(T, ident, box) for _, T, g in pipeline.faces_per_frame(rows, times, 640, 360) for ident, box in g]
    got = sorted(zip(*base[1]))
    assert got == sorted(want)
    FakeDeviceFrame.live = FakeDeviceFrame.peak = 0
    for mode, overlap in (("resident", True), ("stream", True), ("stream", False)):
        other = run_engine(frames, dets, times, shots, mode, overlap=overlap)
        assert other[0] == base[0], (mode, overlap)
        assert other[1] == base[1]
        assert np.array_equal(other[2][0], base[2][0]) and np.array_equal(other[2][1], base[2][1])
    assert FakeDeviceFrame.live == 0                       # every frame the engine staged went back


def test_streaming_holds_shots_in_flight_not_the_video():
    frames, dets, times, shots = make_video(5, n_shots=12, n=20, faces=3)
    FakeDeviceFrame.live = FakeDeviceFrame.peak = 0
    tracks, faces, _, ctx, eng = run_engine(frames, dets, times, shots, "stream")
    assert FakeDeviceFrame.live == 0
    longest = max(i1 - i0 for i0, i1 in engine.split_into_shots(times, shots))
    assert FakeDeviceFrame.peak <= 6 * longest < len(frames)       # one shot each: being read, queued, detected, tracked, extracted (+ slack)
    # track-only runs release a shot's frames when its tracks exist
    FakeDeviceFrame.live = FakeDeviceFrame.peak = 0
    t2, _, _, _, _ = run_engine(frames, dets, times, shots, "stream", extract=False)
    assert t2 == tracks and FakeDeviceFrame.live == 0 and FakeDeviceFrame.peak <= 6 * longest


@pytest.mark.parametrize("window", [1, 5, 17])
def test_windowed_bulk_tracker_starts_equal_whole_shot_ones(window):
    frames, dets, times, shots = make_video(9, n_shots=3, n=40, faces=5, p_miss=0.3, p_false=0.1)
    base = run_engine(frames, dets, times, shots, "resident")
    for overlap in (True, False):
        got = run_engine(frames, dets, times, shots, "resident", overlap=overlap, limit=0, window=window)
        assert got[0] == base[0] and got[1] == base[1]
        assert got[4].stats["windowed_shots"] == 3
        assert got[3].clones == 0                              # windows start the second pass's trackers themselves
    # the trackers alive at any time are those of the windows, not of the shot
    class Counting(FakeContext):
        peak = 0
        def tracker_create_many(self, n, as_array=False):
            r = FakeContext.tracker_create_many(self, n, as_array)
            Counting.peak = max(Counting.peak, len(self.trk))
            return r
    ctx = Counting(frames, dets)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(ctx))
    eng = engine.Engine(ctx, tbd, overlap=False, speculate_limit=0, speculate_window=window)
    job = engine.VideoJob(ctx, 640, 360, frames=frames, times=times, extract=False)
    eng.run(engine.resident_source(job, frames, times, shots, 1), HipTrackers(ctx))
    n_max_shot = max(sum(len(d) for d in dets[i0:i1]) for i0, i1 in engine.split_into_shots(times, shots))
    assert Counting.peak <= 2 * (window + 5) + 12 < n_max_shot


def test_shots_grouped_into_one_set_of_lanes_equal_shot_by_shot():
    """Engine.group (what `--every` uses): the passes of several shots advance in lock-step and share their tracker calls; tracks, faces
    and descriptors are those of the shot-by-shot run, with fewer (larger) on-demand update calls"""
    frames, dets, times, shots = make_video(31, n_shots=7, n=24, faces=4, p_miss=0.6, p_false=0.05)      # many misses: trackers live on
    base = run_engine(frames, dets, times, shots, "resident")
    calls = []
    for group in (1, 3, 8):
        for mode in ("resident", "stream"):
            got = run_engine(frames, dets, times, shots, mode, group=group)
            assert got[0] == base[0] and got[1] == base[1], (group, mode)
            assert np.array_equal(got[2][1], base[2][1])
        calls.append(got[3].update_calls)
    assert calls[0] > calls[1] > calls[2]


@pytest.mark.parametrize("extract_min", [0, 40, 100000, -16])
def test_many_jobs_through_one_engine_run_equal_one_run_each(extract_min, monkeypatch):
    """several videos as jobs of ONE engine run (run_many / the clip farm); extract_min > 0: their faces wait for each other and go through
    the landmark / embedding calls together, across shots and videos (engine.compute_many) -- same rows per video, fewer calls"""
    clips = [make_video(20 + k, n_shots=2, n=18, faces=3) for k in range(4)]
    singles = [run_engine(*c, mode="resident") for c in clips]
    if extract_min < 0:                               # -16: at most 16 faces per call -- batches are cut into pieces, across videos too
        monkeypatch.setattr(engine, "EXTRACT_CALL_MAX", -extract_min)
        extract_min = 100000
    ctx = FakeContext([], [])
    for k, c in enumerate(clips):                     # frame indices restart per clip: key the scripted detections by object
        for f, d in zip(c[0], c[1]):
            ctx.dets_of[id(f)] = d
    ctx.detect_many_orig = ctx.detect_many
    def detect_many(frs, batch, upsample=1, adjust_threshold=0.0, cap=64, arrays=False):
        class W(object):
            def __init__(s, f): s.i = id(f)
        return ctx.detect_many_orig([W(f) for f in frs], batch, upsample, adjust_threshold, cap, arrays)
    ctx.detect_many = detect_many
    landmark_calls = []
    landmarks_orig = ctx.landmarks
    def landmarks(frs, boxes):
        landmark_calls.append(len(boxes))
        return landmarks_orig(frs, boxes)
    ctx.landmarks = landmarks
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(ctx))
    eng = engine.Engine(ctx, tbd, detect_batch_size=7, extract_min=extract_min)
    jobs = [engine.VideoJob(ctx, 640, 360, frames=c[0], times=c[2], key=k) for k, c in enumerate(clips)]
    def source():
        for job, c in zip(jobs, clips):
            for item in engine.resident_source(job, c[0], c[2], c[3], 1):
                yield item
    order = []
    done = eng.run(source(), HipTrackers(ctx), n_shots=8, on_job_final=lambda job: order.append(job.key))
    assert order == [0, 1, 2, 3] and [j.key for j in done] == order
    n_faces = 0
    for job, single in zip(jobs, singles):
        pts, emb = job.ex.finish(computed=True)
        assert job.ex.tracks == single[0]
        assert (job.ex.face_T, job.ex.face_id, job.ex.face_boxes) == single[1]
        assert np.array_equal(pts, single[2][0]) and np.array_equal(emb, single[2][1])
        n_faces += len(pts)
    assert sum(landmark_calls) == n_faces
    if engine.EXTRACT_CALL_MAX == 16:
        assert max(landmark_calls) <= 16 and len(landmark_calls) >= n_faces // 16
    elif extract_min == 100000:
        assert len(landmark_calls) < 12               # everything waited until the GPU thread had no shot left to detect (12 batches exist)
    if extract_min == 40:
        assert len(landmark_calls) < 12               # (12 batches exist: two shots and the end of each of the four videos)


def test_engine_hands_errors_of_either_thread_to_the_caller():
    frames, dets, times, shots = make_video(3, n_shots=3, n=15)
    ctx = FakeContext(frames, dets)
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(ctx))
    def boom(*a, **k):
        raise RuntimeError("detector failed")
    ctx.detect_many = boom
    eng = engine.Engine(ctx, tbd)
    job = engine.VideoJob(ctx, 640, 360, extract=False)
    src = engine.StreamSource(ctx, [(job, [(t, numpy_frame(f)) for t, f in zip(times, frames)], shots, 1, None)])
    with pytest.raises(RuntimeError, match="detector failed"):
        try:
            eng.run(src, HipTrackers(ctx))
        finally:
            src.close()
    # a failing source
    def bad_video():
        yield 0.0, numpy_frame(frames[0])
        raise IOError("decoder died")
    ctx2 = FakeContext(frames, dets)
    tbd2 = TrackingByDetection(detect_func=None, trackers=HipTrackers(ctx2))
    src = engine.StreamSource(ctx2, [(engine.VideoJob(ctx2, 640, 360, extract=False), bad_video(), shots, 1, None)])
    with pytest.raises(IOError, match="decoder died"):
        try:
            engine.Engine(ctx2, tbd2).run(src, HipTrackers(ctx2))
        finally:
            src.close()


def test_fair_lock_serves_in_order_of_arrival():
    """the engine's context lock: a thread that releases it and asks again at once queues behind whoever was already waiting"""
    import threading
    import time
    lock = engine.FairLock()
    order = []
    lock.acquire()
    def waiter():
        with lock:
            order.append("waiter")
    th = threading.Thread(target=waiter)
    th.start()
    while lock.waiting() < 1:
        time.sleep(0.001)
    lock.release()
    with lock:                      # asked for after the waiter: served after it
        order.append("holder again")
    th.join()
    assert order == ["waiter", "holder again"] and lock.waiting() == -1


def test_long_shots_detected_batch_by_batch_equal_whole_shot_calls():
    """a shot of four or more detector batches is detected in one library call per batch, the context lock given up in between (engine._detect);
    the detections, tracks and faces are those of the whole-shot call, the detector is called more often"""
    frames, dets, times, shots = make_video(11, n_shots=3, n=33)
    whole = run_engine(frames, dets, times, shots, "resident", batch=1000)
    pieces = run_engine(frames, dets, times, shots, "resident", batch=7)          # 33+ frames >= 4 x 7: batch by batch
    assert pieces[0] == whole[0] and pieces[1] == whole[1]
    assert np.array_equal(pieces[2][0], whole[2][0]) and np.array_equal(pieces[2][1], whole[2][1])
    assert whole[3].detect_calls == 3 and pieces[3].detect_calls >= 3 * 5


def test_fair_lock_survives_a_waiter_that_gives_up():
    """an exception while waiting for the lock (KeyboardInterrupt) used to leave a ticket nobody serves -- every later acquire
    deadlocked.  A waiter that gives up is skipped; one that gives up at the moment it is served passes the lock on."""
    import threading
    from pyannote_video_amd.engine import FairLock
    lock = FairLock()
    lock.acquire()                                   # ticket 0 holds the lock
    hit = []

    def impatient():
        real_wait = lock._c.wait

        def wait(*a, **k):
            lock._c.wait = real_wait
            raise KeyboardInterrupt()
        lock._c.wait = wait
        try:
            lock.acquire()                           # ticket 1: gives up while waiting
        except KeyboardInterrupt:
            hit.append("interrupted")
    t = threading.Thread(target=impatient); t.start(); t.join()
    assert hit == ["interrupted"]
    got = []
    t2 = threading.Thread(target=lambda: (lock.acquire(), got.append(1), lock.release()))
    t2.start()                                       # ticket 2 waits behind the abandoned ticket 1
    lock.release()
    t2.join(5)
    assert got == [1] and not t2.is_alive()
    with lock:                                       # and the lock is still usable
        assert lock.waiting() == 0


def test_interpreter_tuning_is_reference_counted():
    """two engines at once restored each other's gc / switch-interval settings in the wrong order"""
    import gc
    import sys
    from pyannote_video_amd.engine import _interpreter_tuning as tune
    before = (gc.isenabled(), sys.getswitchinterval())
    with tune:
        assert not gc.isenabled() and sys.getswitchinterval() == pytest.approx(1e-4)
        with tune:
            assert not gc.isenabled()
        assert not gc.isenabled() and sys.getswitchinterval() == pytest.approx(1e-4)     # the inner exit restored nothing
    assert (gc.isenabled(), sys.getswitchinterval()) == (before[0], pytest.approx(before[1]))
