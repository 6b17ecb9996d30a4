"""Test infrastructure: record every call the REFERENCE'S OWN Python makes into `dlib` while its CLI runs (tests/refhost.py executes
/root/reference/scripts/pyannote-face.py verbatim), and replay that call sequence against another dlib look-alike.

/root/reference exists in the build container only, the GPU only on the GPU box: the recorded trace (tests/golden/*/dlib_trace.json:
which object, which method, which frame, which arguments, and what the CPU oracle returned) is the part of that run that travels.
Replayed on the GPU box against pyannote_video_amd.shim it drives the C ABI with exactly the calls, in exactly the order, the
reference makes -- constructors, per-object trackers updated one at a time, rectangles built from tracker positions, landmark objects
handed to the embedder -- and compares every return value (executed without the reference files)."""
import hashlib
import json
import numpy as np


def _key(frame):
    return hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest()


class Recorder(object):
    """module-like wrapper around a dlib look-alike: same surface, every call appended to .trace"""

    def __init__(self, inner, frames):
        self.inner, self.trace = inner, []
        self.index_of = {_key(f): i for i, f in enumerate(frames)}
        self.rectangle, self.drectangle = inner.rectangle, inner.drectangle
        self._n = 0
        rec = self

        class _Detector(object):
            def __init__(s):
                s.o = inner.get_frontal_face_detector()
                rec.trace.append(["detector.new"])

            def __call__(s, rgb, upsample_num_times=0):
                out = s.o(rgb, upsample_num_times)
                rec.trace.append(["detector.call", rec.index_of[_key(rgb)], int(upsample_num_times),
                                  [[r.left(), r.top(), r.right(), r.bottom()] for r in out]])
                return out

        class _Shape(object):
            def __init__(s, path):
                s.o = inner.shape_predictor(path)
                rec.trace.append(["shape_predictor.new"])

            def __call__(s, rgb, rect):
                out = s.o(rgb, rect)
                rec.trace.append(["shape_predictor.call", rec.index_of[_key(rgb)], [rect.left(), rect.top(), rect.right(), rect.bottom()],
                                  [[p.x, p.y] for p in out.parts()]])
                return out

        class _Embed(object):
            def __init__(s, path):
                s.o = inner.face_recognition_model_v1(path)
                rec.trace.append(["face_recognition.new"])

            def compute_face_descriptor(s, rgb, shape, *a):
                out = list(s.o.compute_face_descriptor(rgb, shape, *a))
                rec.trace.append(["face_recognition.call", rec.index_of[_key(rgb)], [[p.x, p.y] for p in shape.parts()], [float(v) for v in out]])
                return out

        class _Tracker(object):
            def __init__(s):
                s.o = inner.correlation_tracker()
                s.k = rec._n
                rec._n += 1
                rec.trace.append(["tracker.new", s.k])

            def start_track(s, rgb, box):
                rec.trace.append(["tracker.start_track", s.k, rec.index_of[_key(rgb)], [box.left(), box.top(), box.right(), box.bottom()]])
                return s.o.start_track(rgb, box)

            def update(s, rgb):
                psr = s.o.update(rgb)
                rec.trace.append(["tracker.update", s.k, rec.index_of[_key(rgb)], float(psr)])
                return psr

            def get_position(s):
                p = s.o.get_position()
                rec.trace.append(["tracker.get_position", s.k, [p.left(), p.top(), p.right(), p.bottom()]])
                return p

        self.get_frontal_face_detector = _Detector
        self.shape_predictor = _Shape
        self.face_recognition_model_v1 = _Embed
        self.correlation_tracker = _Tracker

    def save(self, path, meta):
        with open(path, "w") as f:
            json.dump({"meta": meta, "calls": self.trace}, f, separators=(",", ":"))


def replay(dlib, trace, frames, landmarks_path, embedding_path, embed_tol=1e-4):
    """run the recorded calls against `dlib` (module-like); every return value must equal the recorded one (embeddings within embed_tol).
    Returns the number of calls of each kind."""
    det = sp = emb = last_shape = None
    trackers, counts = {}, {}
    for call in trace:
        kind = call[0]
        counts[kind] = counts.get(kind, 0) + 1
        if kind == "detector.new":
            det = dlib.get_frontal_face_detector()
        elif kind == "shape_predictor.new":
            sp = dlib.shape_predictor(landmarks_path)
        elif kind == "face_recognition.new":
            emb = dlib.face_recognition_model_v1(embedding_path)
        elif kind == "detector.call":
            _, fi, up, want = call
            got = [[r.left(), r.top(), r.right(), r.bottom()] for r in det(frames[fi], up)]
            assert got == want, ("detector", fi, got, want)
        elif kind == "shape_predictor.call":
            _, fi, rect, want = call
            shape = last_shape = sp(frames[fi], dlib.rectangle(*rect))
            assert [[p.x, p.y] for p in shape.parts()] == want, ("landmarks", fi, rect)
        elif kind == "face_recognition.call":
            _, fi, pts, want = call
            # the reference hands the embedder the object the shape predictor has just returned (pyannote-face.py:296-297, face.py:69-76)
            assert [[p.x, p.y] for p in last_shape.parts()] == pts
            got = np.array(list(emb.compute_face_descriptor(frames[fi], last_shape)), np.float64)
            assert np.linalg.norm(got - np.array(want)) <= embed_tol, ("embedding", fi, float(np.linalg.norm(got - np.array(want))))
        elif kind == "tracker.new":
            trackers[call[1]] = dlib.correlation_tracker()
        elif kind == "tracker.start_track":
            _, k, fi, box = call
            trackers[k].start_track(frames[fi], dlib.drectangle(*box))
        elif kind == "tracker.update":
            _, k, fi, want = call
            got = trackers[k].update(frames[fi])
            assert float(got) == want, ("tracker.update", k, fi, got, want)
        elif kind == "tracker.get_position":
            _, k, want = call
            p = trackers[k].get_position()
            assert [p.left(), p.top(), p.right(), p.bottom()] == want, ("tracker.get_position", k)
        else:
            raise ValueError(kind)
    return counts
