"""Test infrastructure: a `dlib` look-alike computed by the CPU ORACLE (oracle/), for running the reference's own Python
(tests/refhost.py) without a GPU.  Same surface as pyannote_video_amd.shim, independent code."""
import numpy as np
from oracle import oracle as O

_MODELS = {}


def configure(detector_model, tables):
    """detector_model: container dict of the HOG detector; tables: models.dsst_tables()"""
    _MODELS["det"] = detector_model
    _MODELS["tabs"] = tables


class point(object):
    def __init__(self, x, y):
        self.x, self.y = int(x), int(y)


class rectangle(object):
    def __init__(self, left, top, right, bottom):
        self._v = (int(left), int(top), int(right), int(bottom))

    def left(self): return self._v[0]
    def top(self): return self._v[1]
    def right(self): return self._v[2]
    def bottom(self): return self._v[3]
    def as_tuple(self): return self._v


class drectangle(object):
    def __init__(self, left, top, right, bottom):
        self._v = (float(left), float(top), float(right), float(bottom))

    def left(self): return self._v[0]
    def top(self): return self._v[1]
    def right(self): return self._v[2]
    def bottom(self): return self._v[3]

    def area(self):
        l, t, r, b = self._v
        return 0.0 if (l > r or t > b) else (r - l) * (b - t)

    def intersect(self, o):
        return drectangle(max(self._v[0], o._v[0]), max(self._v[1], o._v[1]), min(self._v[2], o._v[2]), min(self._v[3], o._v[3]))


class full_object_detection(object):
    def __init__(self, rect, pts):
        self.rect, self._pts = rect, np.asarray(pts, np.int32).reshape(-1, 2)

    def parts(self):
        return [point(x, y) for x, y in self._pts]

    def as_array(self):
        return self._pts


class _Detector(object):
    def __init__(self):
        self._d = O.Detector(_MODELS["det"])

    def __call__(self, rgb, upsample_num_times=0):
        return [rectangle(*b) for b in self._d(rgb, upsample_num_times)]


def get_frontal_face_detector():
    return _Detector()


class shape_predictor(object):
    def __init__(self, path):
        from pyannote_video_amd import models
        self._sp = O.ShapePredictor(models.load_model_file(path, "shape_predictor"))

    def __call__(self, rgb, rect):
        return full_object_detection(rect, self._sp(rgb, rect.as_tuple()))


class face_recognition_model_v1(object):
    def __init__(self, path):
        from pyannote_video_amd import models
        self._e = O.Embedder(models.load_model_file(path, "embedder"))

    def compute_face_descriptor(self, rgb, shape, num_jitters=0):
        return [float(v) for v in self._e(rgb, shape.as_array())]


class correlation_tracker(object):
    def __init__(self):
        self._t = O.Tracker(_MODELS["tabs"])

    def start_track(self, rgb, box):
        self._t.start_track(rgb, (box.left(), box.top(), box.right(), box.bottom()))

    def update(self, rgb):
        return self._t.update(rgb)

    def get_position(self):
        return drectangle(*self._t.get_position())
