"""dlib look-alikes bound to the HIP library: the exact objects the reference's Python calls into.

    reference call site                                   here
    dlib.rectangle / dlib.drectangle                      rectangle / drectangle            (tracking.py:130,167; pyannote-face.py:135,147)
    dlib.get_frontal_face_detector()(rgb, 1)              get_frontal_face_detector()       (face.py:54,66)
    dlib.shape_predictor(path)(rgb, rect).parts()         shape_predictor                   (face.py:58,70; pyannote-face.py:301)
    dlib.face_recognition_model_v1(path)
        .compute_face_descriptor(rgb, shape)              face_recognition_model_v1         (face.py:62,74-75)
    dlib.correlation_tracker() start_track/update/
        get_position                                      correlation_tracker               (tracking.py:203,231,250-251)

Frames may be numpy uint8 HxWx3 arrays (uploaded once and cached by identity) or `DeviceFrame`s already in HBM.
"""
import numpy as np
from . import runtime


class point(object):
    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x, self.y = int(x), int(y)

    def __iter__(self):
        return iter((self.x, self.y))

    def __repr__(self):
        return "point(%d, %d)" % (self.x, self.y)


class rectangle(object):
    """dlib.rectangle: inclusive integer corners, width = right - left + 1"""
    __slots__ = ("_l", "_t", "_r", "_b")

    def __init__(self, left, top, right, bottom):
        self._l, self._t, self._r, self._b = int(left), int(top), int(right), int(bottom)

    def left(self): return self._l
    def top(self): return self._t
    def right(self): return self._r
    def bottom(self): return self._b
    def is_empty(self): return self._l > self._r or self._t > self._b
    def width(self): return 0 if self.is_empty() else self._r - self._l + 1
    def height(self): return 0 if self.is_empty() else self._b - self._t + 1
    def area(self): return self.width() * self.height()

    def intersect(self, o):
        return rectangle(max(self._l, o._l), max(self._t, o._t), min(self._r, o._r), min(self._b, o._b))

    def as_tuple(self):
        return (self._l, self._t, self._r, self._b)

    def __eq__(self, o):
        return isinstance(o, rectangle) and self.as_tuple() == o.as_tuple()

    def __hash__(self):
        return hash(self.as_tuple())

    def __repr__(self):
        return "rectangle(%d,%d,%d,%d)" % self.as_tuple()


class drectangle(object):
    """dlib.drectangle: double corners, width = right - left, empty when left > right or top > bottom"""
    __slots__ = ("_l", "_t", "_r", "_b")

    def __init__(self, left, top, right, bottom):
        self._l, self._t, self._r, self._b = float(left), float(top), float(right), float(bottom)

    def left(self): return self._l
    def top(self): return self._t
    def right(self): return self._r
    def bottom(self): return self._b
    def is_empty(self): return self._l > self._r or self._t > self._b
    def width(self): return 0.0 if self.is_empty() else self._r - self._l
    def height(self): return 0.0 if self.is_empty() else self._b - self._t
    def area(self): return self.width() * self.height()

    def intersect(self, o):
        return drectangle(max(self._l, o._l), max(self._t, o._t), min(self._r, o._r), min(self._b, o._b))

    def as_tuple(self):
        return (self._l, self._t, self._r, self._b)

    def __repr__(self):
        return "drectangle(%r,%r,%r,%r)" % self.as_tuple()


class full_object_detection(object):
    def __init__(self, rect, pts):
        self.rect = rect
        self._pts = np.asarray(pts, np.int32).reshape(-1, 2)

    @property
    def num_parts(self):
        return len(self._pts)

    def part(self, i):
        return point(*self._pts[i])

    def parts(self):
        return [point(x, y) for x, y in self._pts]

    def as_array(self):
        return self._pts


class vector(object):
    """dlib.vector of doubles (what compute_face_descriptor returns): iterable, len 128"""

    def __init__(self, values):
        self._v = np.asarray(values, np.float64)

    def __iter__(self):
        return iter(self._v.tolist())

    def __len__(self):
        return len(self._v)

    def __getitem__(self, i):
        return float(self._v[i])

    def __array__(self, dtype=None, copy=None):
        return self._v if dtype is None else self._v.astype(dtype)


class _Detector(object):
    def __init__(self, ctx=None):
        self._ctx = ctx or runtime.default_context()

    def __call__(self, rgb, upsample_num_times=0):
        boxes, _ = self._ctx.detect(rgb, upsample_num_times)
        return [rectangle(*b) for b in boxes]

    def run(self, rgb, upsample_num_times=0, adjust_threshold=0.0):
        boxes, scores = self._ctx.detect(rgb, upsample_num_times, adjust_threshold)
        return [rectangle(*b) for b in boxes], [float(s) for s in scores], [0] * len(boxes)


def get_frontal_face_detector(ctx=None):
    return _Detector(ctx)


class shape_predictor(object):
    def __init__(self, path, ctx=None):
        self._ctx = ctx or runtime.default_context()
        self._ctx.load_shape_predictor(path)

    def __call__(self, rgb, rect):
        box = rect.as_tuple() if hasattr(rect, "as_tuple") else tuple(rect)
        pts = self._ctx.landmarks([rgb], [box])[0]
        return full_object_detection(rect, pts)


class face_recognition_model_v1(object):
    def __init__(self, path, ctx=None):
        self._ctx = ctx or runtime.default_context()
        self._ctx.load_embedder(path)

    def compute_face_descriptor(self, rgb, shape, num_jitters=0):
        if num_jitters not in (0, 1):
            raise NotImplementedError("num_jitters > 1 is not on the reference's path (face.py:74-75 uses the default)")
        pts = shape.as_array() if hasattr(shape, "as_array") else np.asarray([(p.x, p.y) for p in shape.parts()], np.int32)
        return vector(self._ctx.embed([rgb], [pts])[0])


class correlation_tracker(object):
    """Per-object API of dlib.correlation_tracker; device state is freed when the object dies (tracking.py:123-127 `del`)."""

    def __init__(self, ctx=None):
        self._ctx = ctx or runtime.default_context()
        self._h = self._ctx.tracker_create()

    def __del__(self):
        try:
            if self._h is not None:
                self._ctx.tracker_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def start_track(self, rgb, box):
        b = box.as_tuple() if hasattr(box, "as_tuple") else tuple(box)
        self._ctx.tracker_start_many([self._h], [rgb], [tuple(float(v) for v in b)])

    def update(self, rgb, guess=None):
        if guess is not None:
            raise NotImplementedError("update(img, guess) is not used by the reference (tracking.py:203)")
        psr, _ = self._ctx.tracker_update_many([self._h], [rgb])
        return float(psr[0])

    def get_position(self):
        return drectangle(*self._ctx.tracker_position(self._h))
