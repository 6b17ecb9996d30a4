"""ctypes front-end of the CPU ORACLE (oracle/pvo.h).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package.  PARITY UNPINNED (see pvo.h header): dlib 19.12 / pyannote.algorithms are not available.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def usable_cpus(cap=64):
    """CPUs this process may actually use: min(affinity mask, cgroup v2 / v1 CPU quota), at most `cap`.  A container can show 256
    processors and own 8 of them; an OpenMP team of 256 spinning threads on 8 CPUs ran the oracle-backed GPU tests 9x slower."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, min(n, cap))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libpvo.so")
        if not os.path.exists(path):
            build()
        # before the OpenMP runtime starts: a team sized to the CPUs we own, and threads that sleep instead of spinning between regions
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = C.CDLL(path)
        _LIB.pvo_tracker_new.restype = C.c_void_p
        _LIB.pvo_tracker_update.restype = C.c_double
        _LIB.pvo_det_exp.restype = C.c_double
        _LIB.pvo_det_exp.argtypes = [C.c_double]
        _LIB.pvo_resnet_param_count.restype = C.c_size_t
        _LIB.pvo_shot_dfd.restype = C.c_double
        _LIB.pvo_shot_dfd_from_flow.restype = C.c_double
    return _LIB


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    return img


class _Det(C.Structure):
    _fields_ = [("score", C.c_float), ("filter", C.c_int32), ("level", C.c_int32), ("r", C.c_int32), ("c", C.c_int32),
                ("l", C.c_int32), ("t", C.c_int32), ("rr", C.c_int32), ("b", C.c_int32)]


class _Detector(C.Structure):
    _fields_ = [("n_filters", C.c_int), ("frows", C.c_int), ("fcols", C.c_int), ("cell", C.c_int), ("padding", C.c_int),
                ("win_w", C.c_int), ("win_h", C.c_int), ("min_layer_w", C.c_int), ("min_layer_h", C.c_int),
                ("max_levels", C.c_int), ("nms_iou", C.c_double), ("nms_covered", C.c_double),
                ("w", C.c_void_p), ("thresh", C.c_void_p)]


def resize_bilinear(img, oh, ow):
    img = _u8(img)
    out = np.empty((oh, ow, 3), np.uint8)
    lib().pvo_resize_bilinear_rgb(_p(img), img.shape[0], img.shape[1], _p(out), oh, ow)
    return out


def cv_resize(img, ow, oh):
    """cv2.resize(img, (ow, oh)) (INTER_LINEAR, uint8) restated -- the reference's down-scaled detection frames (video.py:402-403)"""
    img = _u8(img)
    out = np.empty((oh, ow, 3), np.uint8)
    lib().pvo_cv_resize_linear_rgb(_p(img), img.shape[0], img.shape[1], _p(out), oh, ow)
    return out


def pyr_down2(img):
    img = _u8(img)
    oh, ow = C.c_int(), C.c_int()
    lib().pvo_pyr_down2_dims(img.shape[0], img.shape[1], C.byref(oh), C.byref(ow))
    out = np.empty((oh.value, ow.value, 3), np.uint8)
    lib().pvo_pyr_down2_rgb(_p(img), img.shape[0], img.shape[1], _p(out))
    return out


def fhog(img, cell, pad_r, pad_c):
    img = _u8(img)
    fh, fw = C.c_int(), C.c_int()
    lib().pvo_fhog_dims(img.shape[0], img.shape[1], cell, pad_r, pad_c, C.byref(fh), C.byref(fw))
    out = np.zeros((fh.value, fw.value, 32), np.float32)
    lib().pvo_fhog(_p(img), img.shape[0], img.shape[1], cell, pad_r, pad_c, _p(out))
    return out


class Detector(object):
    """dlib.get_frontal_face_detector() restated (reference face.py:54,66)."""

    def __init__(self, model):
        m = model["det.meta"]
        self.w = np.ascontiguousarray(model["det.w"], np.float32)
        self.thresh = np.ascontiguousarray(model["det.thresh"], np.float32)
        nms = model["det.nms"]
        self.s = _Detector(int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5]), int(m[6]), int(m[7]),
                           int(m[8]), int(m[9]), float(nms[0]), float(nms[1]), self.w.ctypes.data, self.thresh.ctypes.data)

    def levels(self, h, w):
        return lib().pvo_detector_levels(h, w, C.byref(self.s))

    def _run(self, fn, rgb, upsample, adjust, cap=1 << 16):
        rgb = _u8(rgb)
        buf = (_Det * cap)()
        n = fn(_p(rgb), rgb.shape[0], rgb.shape[1], int(upsample), C.byref(self.s), C.c_double(adjust), buf, cap)
        return [(buf[i].score, buf[i].filter, buf[i].level, buf[i].r, buf[i].c,
                 (buf[i].l, buf[i].t, buf[i].rr, buf[i].b)) for i in range(n)]

    def detect_raw(self, rgb, upsample=1, adjust=0.0):
        return self._run(lib().pvo_detect_raw, rgb, upsample, adjust)

    def detect(self, rgb, upsample=1, adjust=0.0):
        return self._run(lib().pvo_detect, rgb, upsample, adjust, cap=4096)

    def __call__(self, rgb, upsample=1):
        return [d[5] for d in self.detect(rgb, upsample)]

    def pyramid_level(self, rgb, upsample, level):
        rgb = _u8(rgb)
        oh, ow = C.c_int(), C.c_int()
        lib().pvo_pyramid_level(_p(rgb), rgb.shape[0], rgb.shape[1], upsample, level, None, C.byref(oh), C.byref(ow))
        out = np.empty((oh.value, ow.value, 3), np.uint8)
        lib().pvo_pyramid_level(_p(rgb), rgb.shape[0], rgb.shape[1], upsample, level, _p(out), C.byref(oh), C.byref(ow))
        return out

    def score_level(self, feat, f):
        feat = np.ascontiguousarray(feat, np.float32)
        out = np.zeros(feat.shape[:2], np.float32)
        lib().pvo_score_level(_p(feat), feat.shape[0], feat.shape[1], C.byref(self.s), f, _p(out))
        return out


class _Shape(C.Structure):
    _fields_ = [("n_cascades", C.c_int), ("n_trees", C.c_int), ("n_parts", C.c_int), ("n_pix", C.c_int), ("depth", C.c_int),
                ("initial_shape", C.c_void_p), ("anchor_idx", C.c_void_p), ("deltas", C.c_void_p),
                ("split_idx1", C.c_void_p), ("split_idx2", C.c_void_p), ("split_thresh", C.c_void_p), ("leaves", C.c_void_p)]


class ShapePredictor(object):
    """dlib.shape_predictor(path)(rgb, rect) restated (reference face.py:58,69-70)."""

    def __init__(self, model):
        m = model["sp.meta"]
        self._keep = [np.ascontiguousarray(model[k]) for k in
                      ("sp.initial_shape", "sp.anchor_idx", "sp.deltas", "sp.split_idx1", "sp.split_idx2",
                       "sp.split_thresh", "sp.leaves")]
        self.n_parts = int(m[2])
        self.s = _Shape(int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[4]), *[a.ctypes.data for a in self._keep])

    def __call__(self, rgb, rect):
        rgb = _u8(rgb)
        r = np.asarray(rect, np.int32)
        pts = np.zeros((self.n_parts, 2), np.int32)
        lib().pvo_landmarks(_p(rgb), rgb.shape[0], rgb.shape[1], _p(r), C.byref(self.s), _p(pts))
        return pts


class _Embed(C.Structure):
    _fields_ = [("mean_shape_xy", C.c_void_p), ("chip_size", C.c_int), ("chip_padding", C.c_double),
                ("blob", C.c_void_p), ("blob_len", C.c_size_t)]


class _Chip(C.Structure):
    _fields_ = [("l", C.c_double), ("t", C.c_double), ("r", C.c_double), ("b", C.c_double),
                ("cs", C.c_double), ("sn", C.c_double), ("rows", C.c_int), ("cols", C.c_int)]


def extract_chip(rgb, rect, cs, sn, rows, cols):
    rgb = _u8(rgb)
    d = _Chip(rect[0], rect[1], rect[2], rect[3], cs, sn, rows, cols)
    out = np.zeros((rows, cols, 3), np.uint8)
    lib().pvo_extract_chip_rgb(_p(rgb), rgb.shape[0], rgb.shape[1], C.byref(d), _p(out))
    return out


def transform_image(rgb, m, b, oh, ow):
    rgb = _u8(rgb)
    m = np.asarray(m, np.float64).reshape(4)
    b = np.asarray(b, np.float64).reshape(2)
    out = np.zeros((oh, ow, 3), np.uint8)
    lib().pvo_transform_image_rgb(_p(rgb), rgb.shape[0], rgb.shape[1], _p(m), _p(b), _p(out), oh, ow)
    return out


class Embedder(object):
    """dlib.face_recognition_model_v1(path).compute_face_descriptor(rgb, shape) restated (reference face.py:62,73-76)."""

    def __init__(self, model):
        self.mean = np.ascontiguousarray(model["emb.mean_shape"], np.float32)
        self.blob = np.ascontiguousarray(model["emb.blob"], np.float32)
        assert self.blob.size == lib().pvo_resnet_param_count(), (self.blob.size, lib().pvo_resnet_param_count())
        self.size = int(model["emb.meta"][0])
        self.s = _Embed(self.mean.ctypes.data, self.size, float(model["emb.padding"][0]), self.blob.ctypes.data, self.blob.size)

    def chip_details(self, pts68):
        pts = np.ascontiguousarray(pts68, np.int32)
        d = _Chip()
        lib().pvo_face_chip_details(_p(pts), C.byref(self.s), C.byref(d))
        return (d.l, d.t, d.r, d.b), d.cs, d.sn

    def chip(self, rgb, pts68):
        rect, cs, sn = self.chip_details(pts68)
        return extract_chip(rgb, rect, cs, sn, self.size, self.size)

    def forward(self, chip):
        chip = _u8(chip)
        out = np.zeros(128, np.float32)
        lib().pvo_resnet_forward(_p(chip), C.byref(self.s), _p(out))
        return out

    def __call__(self, rgb, pts68):
        rgb = _u8(rgb)
        pts = np.ascontiguousarray(pts68, np.int32)
        out = np.zeros(128, np.float32)
        lib().pvo_embed(_p(rgb), rgb.shape[0], rgb.shape[1], _p(pts), C.byref(self.s), _p(out))
        return out


class _Tables(C.Structure):
    _fields_ = [("mask64", C.c_void_p), ("mask_scale", C.c_void_p), ("tw64", C.c_void_p), ("tw32", C.c_void_p),
                ("alpha_pow_m16", C.c_double), ("ln_alpha", C.c_double)]


class Tracker(object):
    """dlib.correlation_tracker() restated (reference tracking.py:203,231,250-251)."""

    def __init__(self, tables):
        self._t = {k: (np.ascontiguousarray(v, np.float64) if isinstance(v, np.ndarray) else v) for k, v in tables.items()}
        self._s = _Tables(self._t["mask64"].ctypes.data, self._t["mask_scale"].ctypes.data, self._t["tw64"].ctypes.data,
                          self._t["tw32"].ctypes.data, self._t["alpha_pow_m16"], self._t["ln_alpha"])
        self._h = C.c_void_p(lib().pvo_tracker_new(C.byref(self._s)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().pvo_tracker_free(self._h)
            self._h = None

    def start_track(self, rgb, box):
        rgb = _u8(rgb)
        b = np.asarray(box, np.float64)
        lib().pvo_tracker_start(self._h, _p(rgb), rgb.shape[0], rgb.shape[1], _p(b))

    def update(self, rgb):
        rgb = _u8(rgb)
        return lib().pvo_tracker_update(self._h, _p(rgb), rgb.shape[0], rgb.shape[1])

    def get_position(self):
        b = np.zeros(4, np.float64)
        lib().pvo_tracker_position(self._h, _p(b))
        return tuple(b.tolist())

    def debug_F(self):
        out = np.zeros((32, 64, 64, 2), np.float64)
        lib().pvo_tracker_debug_F(self._h, _p(out))
        return out

    def debug_state(self):
        A = np.zeros((32, 64, 64, 2), np.float64)
        B = np.zeros((64, 64), np.float64)
        lib().pvo_tracker_debug_state(self._h, _p(A), _p(B))
        return A, B


def fft64x64(data, tw64, inverse=False):
    d = np.ascontiguousarray(data, np.float64).copy()
    tw = np.ascontiguousarray(tw64, np.float64)
    lib().pvo_fft64x64(_p(d), _p(tw), int(inverse))
    return d


def det_exp(x):
    return lib().pvo_det_exp(float(x))


def overlap_matrix(a, b, ratio):
    a = np.ascontiguousarray(a, np.float64).reshape(-1, 4)
    b = np.ascontiguousarray(b, np.float64).reshape(-1, 4)
    out = np.zeros((len(a), len(b)), np.float64)
    lib().pvo_overlap_matrix(_p(a), len(a), _p(b), len(b), C.c_double(ratio), _p(out))
    return out


def munkres(cost):
    cost = np.ascontiguousarray(cost, np.float64)
    n = cost.shape[0]
    out = np.zeros(n, np.int32)
    lib().pvo_munkres(_p(cost), n, _p(out))
    return [(i, int(out[i])) for i in range(n)]


def pair_mean_dist(X, row_start):
    X = np.ascontiguousarray(X, np.float64)
    rs = np.ascontiguousarray(row_start, np.int32)
    T = len(rs) - 1
    D = np.zeros((T, T), np.float64)
    lib().pvo_pair_mean_dist(_p(X), X.shape[0], X.shape[1], _p(rs), T, _p(D))
    return D


def hac(D, sizes, threshold):
    D = np.ascontiguousarray(D, np.float64)
    T = D.shape[0]
    sz = np.ascontiguousarray(sizes, np.int32)
    labels = np.zeros(T, np.int32)
    log = np.zeros((max(T - 1, 1), 4), np.float64)
    n = lib().pvo_hac(_p(D), _p(sz), T, C.c_double(threshold), _p(labels), _p(log))
    return labels, log[:n]


# ---- shot boundary detection (structure/shot.py:71-99); PARITY UNPINNED (cv2 absent: cvtColor / resize / Farneback restated)
def shot_tables():
    t = np.zeros(22, np.float32)
    lib().pvo_shot_tables(_p(t))
    return t


def shot_convert(rgb, ow, oh):
    """cv2.resize(cv2.cvtColor(rgb, COLOR_RGB2GRAY), (ow, oh))"""
    rgb = _u8(rgb)
    out = np.zeros((oh, ow), np.uint8)
    lib().pvo_shot_convert(_p(rgb), rgb.shape[0], rgb.shape[1], _p(out), oh, ow)
    return out


def farneback_levels(h, w):
    """coarser pyramid levels OpenCV's Farneback uses for an h x w image (0: a side below 64 pixels)"""
    return int(lib().pvo_farneback_levels(int(h), int(w)))


def farneback(prev, cur, tables=None):
    """cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.1, 0) -> float32 [h, w, 2] (any size: up to three coarser levels)"""
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    assert prev.ndim == 2 and prev.shape == cur.shape
    t = shot_tables() if tables is None else tables
    flow = np.zeros(prev.shape + (2,), np.float32)
    rc = lib().pvo_farneback(_p(prev), _p(cur), prev.shape[0], prev.shape[1], _p(t), _p(flow))
    if rc != 0:
        raise RuntimeError("pvo_farneback failed (%d)" % rc)
    return flow


farneback_small = farneback          # (the name of the single-level restatement of round 3)


def shot_dfd(prev, cur, tables=None):
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    t = shot_tables() if tables is None else tables
    return float(lib().pvo_shot_dfd(_p(prev), _p(cur), prev.shape[0], prev.shape[1], _p(t)))
