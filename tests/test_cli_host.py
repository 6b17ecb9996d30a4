"""CPU: the `extract` verb's own work (cli.extract: a reader thread feeding frames that carry faces, batches through ONE landmark +
descriptor call, a writer thread formatting and writing batch i while batch i + 1 is computed) on a scripted context: the two files must
hold, line for line, what a plain loop over getFaceGenerator's pairing (pipeline.faces_per_frame, pyannote-face.py:121-175, 287-311)
writes, whatever the batch size; an error on any of the three threads reaches the caller."""
import numpy as np
import pytest

from pyannote_video_amd import cli, formats, pipeline


class ScriptVideo(object):
    def __init__(self, n, w=64, h=48, fps=25.0):
        self.n, self.frame_size, self.size, self.frame_rate = n, (w, h), (w, h), fps

    def __len__(self):
        return self.n

    def __iter__(self):
        w, h = self.frame_size
        for i in range(self.n):
            a = np.zeros((h, w, 3), np.uint8)
            a[0, 0, 0], a[0, 0, 1] = i // 256, i % 256
            yield i / self.frame_rate, a


class Dev(object):
    def __init__(self, i, shape):
        self.i, self.shape, self.released = i, shape, False

    def release(self):
        assert not self.released
        self.released = True


class Ring(object):
    def __init__(self, log):
        self.log = log

    def push(self, rgb):
        d = Dev(int(rgb[0, 0, 0]) * 256 + int(rgb[0, 0, 1]), rgb.shape)
        self.log.append(d)
        return d

    def close(self):
        pass


class ScriptContext(object):
    def __init__(self, fail_at=None):
        self.calls, self.frames, self.fail_at = [], [], fail_at

    def load_shape_predictor(self, path):
        pass

    def load_embedder(self, path):
        pass

    def ingest_ring(self, h, w, depth=16):
        return Ring(self.frames)

    def landmarks_embed(self, frames, boxes):
        self.calls.append(len(boxes))
        if self.fail_at is not None and len(self.calls) > self.fail_at:
            raise RuntimeError("scripted device error")
        pts = np.zeros((len(boxes), 68, 2), np.int32)
        emb = np.zeros((len(boxes), 128), np.float32)
        for k, (f, b) in enumerate(zip(frames, boxes)):
            assert not f.released, "face computed on a released frame"
            pts[k, :, 0] = b[0] + np.arange(68) + f.i
            pts[k, :, 1] = b[1] + 2 * np.arange(68)
            emb[k] = np.sin(0.01 * (f.i + b[0] + b[1]) + 0.1 * np.arange(128)) * 0.1
        return pts, emb


def make_tracks(n_frames, fps=25.0, seed=0):
    rng = np.random.default_rng(seed)
    tracks = []
    for k in range(5):
        a = int(rng.integers(0, n_frames // 2))
        b = int(rng.integers(a + 3, n_frames))
        x, y = rng.uniform(0.1, 0.5, 2)
        tracks.append([(i / fps, (x + 0.001 * i, y, x + 0.3, y + 0.3), "forward" if i > a else "detection") for i in range(a, b)])
    return tracks


def expected_files(video, track_path):
    rows = formats.read_tracks(track_path)
    w, h = video.frame_size
    times = [i / video.frame_rate for i in range(len(video))]
    ctx = ScriptContext()
    lm, em = [], []
    for fi, T, g in pipeline.faces_per_frame(rows, times, w, h):
        f = Dev(fi, (h, w, 3))
        for ident, box in g:
            pts, emb = ctx.landmarks_embed([f], [box])
            lm.append(formats.landmark_rows([T], [ident], pts, w, h))
            em.append(formats.embedding_rows([T], [ident], emb))
    return b"".join(lm), b"".join(em)


@pytest.mark.parametrize("batch", [1, 7, 2048])
def test_extract_verb_files_equal_the_plain_loop(tmp_path, batch):
    video = ScriptVideo(60)
    tp = str(tmp_path / "track.txt")
    formats.write_tracks(tp, make_tracks(60))
    want_lm, want_em = expected_files(video, tp)
    assert want_lm.count(b"\n") > 50
    ctx = ScriptContext()
    lp, ep = str(tmp_path / "landmarks.txt"), str(tmp_path / "embedding.txt")
    cli.extract(video, "unused", "unused", tp, lp, ep, ctx=ctx, batch=batch, ahead=4)
    assert open(lp, "rb").read() == want_lm and open(ep, "rb").read() == want_em
    assert all(d.released for d in ctx.frames) and len(ctx.frames) > 0          # frames the verb staged itself went back
    if batch == 2048:
        assert len(ctx.calls) == 1
    if batch == 1:
        assert len(ctx.calls) >= want_lm.count(b"\n") // 5                      # (a frame's faces stay together)


def test_extract_verb_hands_on_errors_of_its_threads(tmp_path):
    video = ScriptVideo(60)
    tp = str(tmp_path / "track.txt")
    formats.write_tracks(tp, make_tracks(60))
    lp, ep = str(tmp_path / "landmarks.txt"), str(tmp_path / "embedding.txt")
    with pytest.raises(RuntimeError, match="scripted device error"):               # the computing thread
        cli.extract(video, "unused", "unused", tp, lp, ep, ctx=ScriptContext(fail_at=1), batch=7, ahead=4)

    class BadVideo(ScriptVideo):                                                    # the reader thread
        def __iter__(self):
            for k, item in enumerate(ScriptVideo.__iter__(self)):
                if k == 20:
                    raise IOError("scripted decode error")
                yield item
    with pytest.raises(IOError, match="scripted decode error"):
        cli.extract(BadVideo(60), "unused", "unused", tp, lp, ep, ctx=ScriptContext(), batch=7, ahead=4)
    orig = formats.embedding_rows                                                   # the writer thread
    try:
        def boom(*a, **k):
            raise ValueError("scripted formatter error")
        formats.embedding_rows = boom
        with pytest.raises(ValueError, match="scripted formatter error"):
            cli.extract(video, "unused", "unused", tp, lp, ep, ctx=ScriptContext(), batch=7, ahead=4)
    finally:
        formats.embedding_rows = orig


# ---- `track` and `process` on the streaming engine, scripted context (detections, trackers, landmark / descriptor stand-ins of test_engine) ----
def _engine_fixture(seed=5):
    from tests.test_engine import FakeContext, make_video, numpy_frame, run_engine
    frames, dets, times, shots = make_video(seed, n_shots=3, n=20)

    class Ctx(FakeContext):
        def load_shape_predictor(self, path):
            pass

        def load_embedder(self, path):
            pass

        def landmarks_embed(self, fr, boxes):             # (the `extract` verb's one call; the engine may use either form)
            pts = self.landmarks(fr, boxes)
            return pts, self.embed(fr, pts)

    class Video(object):
        frame_rate, size, frame_size = 25.0, (640, 360), (640, 360)

        def __len__(self):
            return len(frames)

        def __iter__(self):
            for t, f in zip(times, frames):
                yield t, numpy_frame(f)

    want = run_engine(frames, dets, times, shots, "resident")
    return Ctx(frames, dets), Video(), shots, want


def test_track_verb_writes_the_engine_s_tracks(tmp_path):
    ctx, video, shots, want = _engine_fixture()
    out = str(tmp_path / "track.txt")
    cli.track(video, shots, out, ctx=ctx)
    lines = [l for i, trk in enumerate(want[0]) for l in formats.track_lines(i, trk)]
    assert len(lines) > 100 and open(out).readlines() == lines
    assert not ctx.trk


def test_process_verb_writes_what_track_then_extract_write(tmp_path):
    """one pass (cli.process) == the two verbs one after the other, file for file; and == the engine's own arrays"""
    ctx, video, shots, want = _engine_fixture()
    p = {k: str(tmp_path / (k + ".txt")) for k in ("t1", "l1", "e1", "t2", "l2", "e2")}
    cli.process(video, shots, "unused", "unused", p["t1"], p["l1"], p["e1"], ctx=ctx)
    ctx2, video2, shots2, _ = _engine_fixture()
    cli.track(video2, shots2, p["t2"], ctx=ctx2)
    cli.extract(video2, "unused", "unused", p["t2"], p["l2"], p["e2"], ctx=ctx2, batch=16)
    for a, b in (("t1", "t2"), ("l1", "l2"), ("e1", "e2")):
        assert open(p[a], "rb").read() == open(p[b], "rb").read(), a
    face_T, face_id, _ = want[1]
    pts, emb = want[2]
    assert open(p["l1"], "rb").read() == formats.landmark_rows(face_T, face_id, pts, 640, 360)
    assert open(p["e1"], "rb").read() == formats.embedding_rows(face_T, face_id, emb)


def test_pipeline_run_many_equals_run_per_clip():
    """FacePipeline.run_many (the clip farm, configs[3]): every clip's result dictionary == FacePipeline.run on that clip alone, although the
    farm extracts the faces of several clips in one call (Engine.extract_min)"""
    from tests.test_engine import FakeContext, make_video
    clips = [make_video(40 + k, n_shots=2, n=16, faces=3) for k in range(3)]

    class Ctx(FakeContext):
        def __init__(self):
            FakeContext.__init__(self, [], [])
            self.embed_calls = []

        def load_shape_predictor(self, path):
            pass

        def load_embedder(self, path):
            pass

        def detect_many(self, frs, batch, upsample=1, adjust_threshold=0.0, cap=64, arrays=False):
            class W(object):
                def __init__(s, f):
                    s.i = id(f)
            return FakeContext.detect_many(self, [W(f) for f in frs], batch, upsample, adjust_threshold, cap, arrays)

        def embed(self, frames, pts):
            self.embed_calls.append(len(pts))
            return FakeContext.embed(self, frames, pts)

    def fresh():
        ctx = Ctx()
        for c in clips:
            for f, d in zip(c[0], c[1]):
                ctx.dets_of[id(f)] = d
        return ctx, pipeline.FacePipeline(ctx, "unused", "unused", detect_batch_size=7)

    ctx, pipe = fresh()
    singles = [pipe.run(c[0], c[2], 25.0, c[3], cluster=False) for c in clips]
    ctx, pipe = fresh()
    seen = []
    farm = pipe.run_many([dict(frames=c[0], times=c[2], frame_rate=25.0, shots=c[3]) for c in clips], cluster=False,
                         on_result=lambda k, r: seen.append(k))
    assert seen == [0, 1, 2]                    # (how many calls the farm saves depends on timing: tests/test_engine.py pins the mechanism)
    for a, b in zip(farm, singles):
        assert a["tracks"] == b["tracks"] and a["track_rows"] == b["track_rows"]
        assert np.array_equal(a["face_T"], b["face_T"]) and np.array_equal(a["face_id"], b["face_id"]) and a["face_boxes"] == b["face_boxes"]
        assert np.array_equal(a["landmarks"], b["landmarks"]) and np.array_equal(a["embeddings"], b["embeddings"]) and np.array_equal(a["X"], b["X"])
        assert len(a["face_T"]) > 20


def test_tracking_call_streams_through_the_engine():
    """TrackingByDetection.__call__(video, segmentation) -- the reference's entry point (tracking.py:374-434) -- with the GPU tracker backend:
    the video goes through the streaming engine in a worker thread, the tracks come out shot after shot, equal to the engine's own"""
    from tests.test_engine import FakeContext, make_video, numpy_frame, run_engine
    from pyannote_video_amd.tracking_by_detection import TrackingByDetection, HipTrackers
    from pyannote_video_amd._core import Segment
    frames, dets, times, shots = make_video(6, n_shots=3, n=20)
    ctx = FakeContext(frames, dets)

    class Video(object):
        frame_rate, size, frame_size = 25.0, (640, 360), (640, 360)

        def __len__(self):
            return len(frames)

        def __iter__(self):
            for t, f in zip(times, frames):
                yield t, numpy_frame(f)

    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(ctx))
    got = list(tbd(Video(), [Segment(a, b) for a, b in shots]))
    want = run_engine(frames, dets, times, shots, "resident", extract=False)[0]
    assert len(got) > 5 and got == want
    assert not ctx.trk

    class BadVideo(Video):                      # an error in the reader thread reaches the caller of the generator
        def __iter__(self):
            for k, item in enumerate(Video.__iter__(self)):
                if k == 30:
                    raise IOError("scripted decode error")
                yield item
    tbd = TrackingByDetection(detect_func=None, track_min_overlap_ratio=0.5, track_max_gap=1.0, trackers=HipTrackers(FakeContext(frames, dets)))
    with pytest.raises(IOError, match="scripted decode error"):
        list(tbd(BadVideo(), [Segment(a, b) for a, b in shots]))
