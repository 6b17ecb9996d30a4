"""`Context`: one GPU, one HIP stream, the loaded models and the frames staged in HBM.  Thin Python over the C ABI."""
import contextlib
import ctypes as C
import threading
import itertools
import operator
import numpy as np
from . import _lib
from ._lib import check, ptr, handles
from . import models as _models


class DeviceFrame(object):
    """A frame resident in HBM (uint8 RGB HWC).  `keep` pins whatever owns the memory (e.g. a torch tensor)."""
    __slots__ = ("ctx", "handle", "height", "width", "keep", "transient", "__weakref__")

    def __init__(self, ctx, handle, height, width, keep=None, transient=False):
        self.ctx, self.handle, self.height, self.width, self.keep = ctx, handle, height, width, keep
        self.transient = transient      # a streaming source made it for one pass: the engine releases it when `extract` has passed it

    @property
    def shape(self):
        return (self.height, self.width, 3)

    def release(self):
        if self.handle is not None and self.ctx._h is not None:
            _lib.lib().pvf_frame_release(self.ctx._h, self.handle)
        self.handle = None
        self.keep = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class IngestRing(object):
    """Pinned host slots of one frame size + a copy stream (pvf_ingest_*).  `slot()` hands out the next slot as a numpy view for the
    decoder to write into; `submit()` queues its upload and returns a DeviceFrame at once -- kernels that read the frame wait for
    the copy on the device, so uploads run beside the compute stream."""

    def __init__(self, ctx, height, width, depth=8):
        self.ctx, self.h, self.w, self.depth = ctx, int(height), int(width), int(depth)
        r = C.c_uint64(0)
        check(ctx._l.pvf_ingest_create(ctx._h, self.h, self.w, self.depth, C.byref(r)))
        self._r = r.value
        self._cur = None

    def slot(self):
        s, p = C.c_int32(0), C.c_void_p(0)
        check(self.ctx._l.pvf_ingest_acquire(self.ctx._h, self._r, C.byref(s), C.byref(p)))
        self._cur = s.value
        buf = (C.c_uint8 * (self.h * self.w * 3)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(self.h, self.w, 3)

    def submit(self):
        h = C.c_uint64(0)
        check(self.ctx._l.pvf_ingest_submit(self.ctx._h, self._r, self._cur, C.byref(h)))
        return DeviceFrame(self.ctx, h.value, self.h, self.w)

    def wait(self):
        check(self.ctx._l.pvf_ingest_wait(self.ctx._h, self._r))

    def push(self, rgb):
        """copy a decoded frame into the next slot and queue its upload (a decoder would write into slot() directly)"""
        np.copyto(self.slot(), rgb)
        return self.submit()

    def close(self):
        if self._r is not None and self.ctx._h is not None:
            self.ctx._l.pvf_ingest_destroy(self.ctx._h, self._r)
        self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _boxes_and_scores(out, scores, counts):
    """[( [(l, t, r, b) Python ints], float32 scores )] per frame.  One tolist() for the whole batch: indexing numpy rows element by
    element cost 3 ms per 250-frame shot, during which the GPU had nothing queued."""
    cnt = counts.tolist()
    rows = out[:, :max(cnt, default=0)].tolist()           # only the filled slots
    return [([tuple(b) for b in rows[i][:cnt[i]]], scores[i, :cnt[i]].copy()) for i in range(len(cnt))]


class DeviceRows(object):
    """rows in device memory handed to the library by address: n rows, `stride` bytes apart; `keep` = whatever owns the memory
    (a torch tensor, a buffer of another library), kept alive as long as this object"""

    def __init__(self, ptr, n, stride, keep=None):
        self.ptr, self.n, self.stride, self.keep = int(ptr), int(n), int(stride), keep

    @classmethod
    def of_tensor(cls, t):
        """a 2-D contiguous torch tensor on the device"""
        assert t.dim() == 2 and t.is_contiguous()
        return cls(t.data_ptr(), t.shape[0], t.shape[1] * t.element_size(), keep=t)


_handle_of = operator.attrgetter("handle")


def _rects(boxes, n):
    """pvf_rect_i32[n] from n boxes: int() of each coordinate.  Lists of integer 4-tuples (what `extract` hands over, thousands per call)
    go through one iterator pass -- a quarter of the time numpy takes to parse a list of tuples, on a thread the GPU is waiting for."""
    if not isinstance(boxes, np.ndarray):
        try:
            r = np.fromiter(itertools.chain.from_iterable(boxes), dtype=np.int64)
            if r.size == 4 * n:
                return r.astype(np.int32).reshape(n, 4)
        except (TypeError, ValueError):
            pass                                            # floats, arrays, ragged input: the general form below decides
    return np.ascontiguousarray(np.asarray(boxes).astype(np.int64).astype(np.int32)).reshape(n, 4)


class Context(object):
    _stage_mu = threading.RLock()      # (instances make their own in __init__; this one serves objects built without it, e.g. test doubles)

    def __init__(self, device=0, detector=_models.DEFAULT_DETECTOR, landmarks=None, embedding=None, priority=0):
        self._h = None
        l = _lib.lib()
        h = C.c_uint64(0)
        check(l.pvf_ctx_create_prio(int(device), int(priority), C.byref(h)))
        self._h = h.value
        self.device = int(device)
        self._l = l
        self._staged = {}      # (id, data ptr) -> DeviceFrame for numpy frames passed through the dlib-like API
        self._staged_order = []
        self.stage_capacity = 1024
        self._hold = 0         # > 0 while a call is collecting frame handles: nothing staged may be evicted until it has run
        self._stage_mu = threading.RLock()   # the staging cache is used from the detector thread and the tracker / extraction threads
        self._tables = False
        self._models = {}
        if detector:
            self.load_detector(detector)
        if landmarks:
            self.load_shape_predictor(landmarks)
        if embedding:
            self.load_embedder(embedding)

    def close(self):
        if self._h is not None:
            for f in list(self._staged.values()):
                f.release()
            self._staged.clear()
            self._l.pvf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._l.pvf_sync(self._h))

    # ---- models
    def load_detector(self, path):
        check(self._l.pvf_load_detector(self._h, str(path).encode()))

    def _model_key(self, path):
        import os
        try:
            st = os.stat(str(path))
        except OSError:
            return object()          # never equal: the loader reports the missing file
        return (os.path.abspath(str(path)), st.st_size, st.st_mtime_ns)

    def load_shape_predictor(self, path):
        """(a model file that is already loaded -- same path, size and modification time -- is not read again: every CLI verb and
        every FacePipeline names its models)"""
        key = self._model_key(path)
        if self._models.get("sp") != key:
            check(self._l.pvf_load_shape_predictor(self._h, str(path).encode()))
            self._models["sp"] = key

    def load_embedder(self, path):
        key = self._model_key(path)
        if self._models.get("emb") != key:
            check(self._l.pvf_load_embedder(self._h, str(path).encode()))
            self._models["emb"] = key

    def ensure_tracker_tables(self):
        if not self._tables:
            t = _models.dsst_tables()
            self._tab_keep = t
            check(self._l.pvf_set_tracker_tables(self._h, ptr(t["mask64"]), ptr(t["mask_scale"]), ptr(t["tw64"]), ptr(t["tw32"]),
                                                 t["alpha_pow_m16"], t["ln_alpha"]))
            self._tables = True

    # ---- frames
    def upload(self, rgb):
        rgb = np.asarray(rgb)
        if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
            raise TypeError("frames must be uint8 arrays of shape (H, W, 3)")   # dlib raises on unsupported arrays too
        if not rgb.flags["C_CONTIGUOUS"]:
            rgb = np.ascontiguousarray(rgb)
        h = C.c_uint64(0)
        check(self._l.pvf_frame_upload(self._h, ptr(rgb), rgb.shape[0], rgb.shape[1], rgb.strides[0], C.byref(h)))
        return DeviceFrame(self, h.value, rgb.shape[0], rgb.shape[1])

    def upload_device(self, data_ptr, height, width, transient=False):
        """copy of a frame that already lies in HBM (a decoder's output surface) into a buffer of the library's own"""
        h = C.c_uint64(0)
        check(self._l.pvf_frame_upload(self._h, C.c_void_p(int(data_ptr)), int(height), int(width), int(width) * 3, C.byref(h)))
        return DeviceFrame(self, h.value, int(height), int(width), transient=transient)

    def mem_info(self):
        """(free, total) bytes of device memory as the driver sees them"""
        f, t = C.c_int64(0), C.c_int64(0)
        check(self._l.pvf_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def pool_trim(self, keep_bytes=0):
        """give pooled frame buffers beyond keep_bytes back to the allocator; returns what the pool still holds"""
        n = C.c_int64(0)
        check(self._l.pvf_frame_pool_trim(self._h, int(keep_bytes), C.byref(n)))
        return n.value

    def wrap_device(self, data_ptr, height, width, keep=None):
        h = C.c_uint64(0)
        check(self._l.pvf_frame_wrap_device(self._h, C.c_void_p(int(data_ptr)), int(height), int(width), C.byref(h)))
        return DeviceFrame(self, h.value, int(height), int(width), keep)

    def share(self, frame):
        """DeviceFrame of THIS context for a frame staged in another context on the same GPU (no copy)"""
        if frame.ctx is self:
            return frame
        p = C.c_void_p(0)
        check(self._l.pvf_frame_device_ptr(frame.ctx._h, frame.handle, C.byref(p)))
        return self.wrap_device(p.value, frame.height, frame.width, keep=frame)

    def wrap_torch(self, t):
        """t: torch.uint8 CUDA tensor [H, W, 3], contiguous"""
        assert t.is_cuda and t.is_contiguous() and t.dim() == 3 and t.shape[2] == 3 and t.element_size() == 1
        return self.wrap_device(t.data_ptr(), t.shape[0], t.shape[1], keep=t)

    def resize(self, frame, width, height):
        """cv2.resize(frame, (width, height)) on the device (reference video.py:402-403): a new DeviceFrame; the source stays resident"""
        f = self.stage(frame)
        h = C.c_uint64(0)
        check(self._l.pvf_frame_resize(self._h, f.handle, int(width), int(height), C.byref(h)))
        return DeviceFrame(self, h.value, int(height), int(width))

    def ingest_ring(self, height, width, depth=8):
        return IngestRing(self, height, width, depth)

    def stage(self, rgb):
        """DeviceFrame for whatever the caller holds: DeviceFrame (as is) or numpy array (uploaded once, cached by identity)."""
        if isinstance(rgb, DeviceFrame):
            return rgb
        key = (id(rgb), rgb.__array_interface__["data"][0] if hasattr(rgb, "__array_interface__") else 0)
        with self._stage_mu:
            f = self._staged.get(key)
            if f is None or f.keep is not rgb:
                f = self.upload(rgb)
                f.keep = rgb   # keeps the id stable while cached
                self._staged[key] = f
                self._staged_order.append(key)
                if not self._hold:
                    self._trim()
            return f

    def _trim(self):
        with self._stage_mu:
            while len(self._staged_order) > self.stage_capacity:
                old = self._staged_order.pop(0)
                g = self._staged.pop(old, None)
                if g is not None:
                    g.release()

    @contextlib.contextmanager
    def _staging(self):
        """Frames staged inside the block stay resident until the block ends: one call may reference more distinct numpy
        frames than the cache holds (a 4096-tracker batch of a long shot), and a handle released before the C call runs is
        an 'unknown frame handle'.  The cache is trimmed back to its capacity afterwards."""
        with self._stage_mu:
            self._hold += 1
        try:
            yield
        finally:
            with self._stage_mu:
                self._hold -= 1
                if not self._hold:
                    self._trim()

    def _handles(self, frames):
        if isinstance(frames, np.ndarray) and frames.dtype == np.uint64:     # handles the caller looked up before (frame_handles)
            return np.ascontiguousarray(frames)
        try:                                                # frames already on the device (the engine's case): one pass at C speed
            return np.fromiter(map(_handle_of, frames), dtype=np.uint64, count=len(frames))
        except AttributeError:
            return handles([self.stage(f).handle for f in frames])

    def frame_handles(self, frames):
        """uint64 array of the staged frames' handles: look them up once, index the array for every batched call"""
        return self._handles(frames)

    def unstage_all(self):
        for f in self._staged.values():
            f.release()
        self._staged.clear()
        self._staged_order = []

    # ---- S1
    def detect_batch(self, frames, upsample=1, adjust_threshold=0.0, cap=256):
        n = len(frames)
        out = np.zeros((n, cap, 4), np.int32)
        scores = np.zeros((n, cap), np.float32)
        counts = np.zeros(n, np.int32)
        with self._staging():
            hs = self._handles(frames)
            check(self._l.pvf_detect_batch(self._h, ptr(hs), n, int(upsample), float(adjust_threshold), ptr(out), ptr(scores), ptr(counts), cap))
        if int(counts.max(initial=0)) >= cap:     # a frame filled its slots: repeat with room for every detection (like detect_many)
            return self.detect_batch(frames, upsample, adjust_threshold, cap * 8)
        return _boxes_and_scores(out, scores, counts)

    def detect_many(self, frames, batch, upsample=1, adjust_threshold=0.0, cap=64, arrays=False):
        """any number of frames of one size, `batch` at a time, host post-processing overlapped with the next batch's kernels.
        arrays=True: (boxes int32 [n, slots, 4], scores float32 [n, slots], counts int32 [n]) instead of Python lists -- a caller that
        feeds the boxes straight back to the GPU converts them to Python objects when (and where) it has the time"""
        n = len(frames)
        out = np.zeros((n, cap, 4), np.int32)
        scores = np.zeros((n, cap), np.float32)
        counts = np.zeros(n, np.int32)
        with self._staging():
            hs = self._handles(frames)
            check(self._l.pvf_detect_many(self._h, ptr(hs), n, int(batch), int(upsample), float(adjust_threshold), ptr(out), ptr(scores), ptr(counts), cap))
        if int(counts.max(initial=0)) >= cap:     # a frame filled its slots: repeat with room for every detection
            return self.detect_many(frames, batch, upsample, adjust_threshold, cap * 8, arrays)
        if arrays:
            m = int(counts.max(initial=0))
            return out[:, :m], scores[:, :m], counts
        return _boxes_and_scores(out, scores, counts)

    def detect(self, frame, upsample=1, adjust_threshold=0.0):
        return self.detect_batch([frame], upsample, adjust_threshold)[0]

    def detect_raw(self, frame, upsample=1, adjust_threshold=0.0, cap=65536):
        f = self.stage(frame)
        scores = np.zeros(cap, np.float32)
        meta = np.zeros((cap, 8), np.int32)
        n = C.c_int32(0)
        check(self._l.pvf_debug_detect_raw(self._h, f.handle, int(upsample), float(adjust_threshold), ptr(scores), ptr(meta), cap, C.byref(n)))
        k = min(n.value, cap)
        return [(float(scores[i]), int(meta[i, 0]), int(meta[i, 1]), int(meta[i, 2]), int(meta[i, 3]),
                 tuple(int(v) for v in meta[i, 4:8])) for i in range(k)]

    def detect_raw_many(self, frames, batch, upsample=1, adjust_threshold=0.0, cap=1 << 16):
        """the scanner's candidates BEFORE non-maximum suppression through the batched path (screening pass included): per frame an int32
        [n, 5] table (level, filter, row, column, score bits) in the detector's canonical order"""
        n = len(frames)
        counts = np.zeros(n, np.int32)
        total = C.c_int64(0)
        with self._staging():
            hs = self._handles(frames)
            while True:
                rows = np.zeros((cap, 5), np.int32)
                check(self._l.pvf_debug_detect_raw_many(self._h, ptr(hs), n, int(batch), int(upsample), float(adjust_threshold), ptr(counts), ptr(rows), cap, C.byref(total)))
                if total.value <= cap:
                    break
                cap = int(total.value)
        off = np.concatenate([[0], np.cumsum(counts)])
        return [rows[off[i]:off[i + 1]] for i in range(n)]

    def pyramid_batch(self, frames, upsample=1):
        """the image pyramids of a batch and nothing else (measurement: the detector's resize chain alone)"""
        with self._staging():
            hs = self._handles(frames)
            check(self._l.pvf_debug_pyramid_batch(self._h, ptr(hs), len(frames), int(upsample)))

    def pyramid_level(self, frame, upsample, level):
        f = self.stage(frame)
        oh, ow = C.c_int32(0), C.c_int32(0)
        check(self._l.pvf_debug_pyramid_level(self._h, f.handle, upsample, level, None, C.byref(oh), C.byref(ow)))
        out = np.zeros((oh.value, ow.value, 3), np.uint8)
        check(self._l.pvf_debug_pyramid_level(self._h, f.handle, upsample, level, ptr(out), C.byref(oh), C.byref(ow)))
        return out

    def level_features(self, frame, upsample, level):
        """FHOG features [fh][fw][32] of one pyramid level as the batched detector computes them (parity tests)."""
        f = self.stage(frame)
        fh, fw = C.c_int32(0), C.c_int32(0)
        check(self._l.pvf_debug_level_features(self._h, f.handle, upsample, level, None, C.byref(fh), C.byref(fw)))
        out = np.zeros((fh.value, fw.value, 32), np.float32)
        if out.size:
            check(self._l.pvf_debug_level_features(self._h, f.handle, upsample, level, ptr(out), C.byref(fh), C.byref(fw)))
        return out

    def fhog(self, img, cell, pad_r, pad_c):
        img = np.ascontiguousarray(img, np.uint8)
        fh, fw = C.c_int32(0), C.c_int32(0)
        check(self._l.pvf_debug_fhog(self._h, ptr(img), img.shape[0], img.shape[1], cell, pad_r, pad_c, None, C.byref(fh), C.byref(fw)))
        out = np.zeros((fh.value, fw.value, 32), np.float32)
        check(self._l.pvf_debug_fhog(self._h, ptr(img), img.shape[0], img.shape[1], cell, pad_r, pad_c, ptr(out), C.byref(fh), C.byref(fw)))
        return out

    # ---- S2
    def tracker_create(self):
        self.ensure_tracker_tables()
        h = C.c_uint64(0)
        check(self._l.pvf_tracker_create(self._h, C.byref(h)))
        return h.value

    def tracker_create_many(self, n, as_array=False):
        self.ensure_tracker_tables()
        out = np.zeros(int(n), np.uint64)
        if n:
            check(self._l.pvf_tracker_create_many(self._h, int(n), ptr(out)))
        return out if as_array else out.tolist()

    def tracker_clone_many(self, trks, as_array=False):
        out = np.zeros(len(trks), np.uint64)
        if len(trks):
            check(self._l.pvf_tracker_clone_many(self._h, ptr(handles(trks)), len(trks), ptr(out)))
        return out if as_array else out.tolist()

    def tracker_destroy_many(self, trks):
        if self._h is not None and len(trks):
            check(self._l.pvf_tracker_destroy_many(self._h, ptr(handles(trks)), len(trks)))

    def tracker_destroy(self, trk):
        if self._h is not None:
            check(self._l.pvf_tracker_destroy(self._h, trk))

    def tracker_start_many(self, trks, frames, boxes):
        if not len(trks):
            return
        b = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
        with self._staging():
            check(self._l.pvf_tracker_start_many(self._h, ptr(handles(trks)), ptr(self._handles(frames)), ptr(b), len(trks)))

    def tracker_update_many(self, trks, frames, defer=False):
        """defer=True: confidence and position only, the filter update is left to tracker_commit_many (same frames)"""
        n = len(trks)
        if not n:
            return np.zeros(0), np.zeros((0, 4))
        psr = np.zeros(n, np.float64)
        boxes = np.zeros((n, 4), np.float64)
        fn = self._l.pvf_tracker_update_many_deferred if defer else self._l.pvf_tracker_update_many
        with self._staging():
            check(fn(self._h, ptr(handles(trks)), ptr(self._handles(frames)), n, ptr(psr), ptr(boxes)))
        return psr, boxes

    def tracker_commit_many(self, trks, frames):
        n = len(trks)
        if n:
            with self._staging():
                check(self._l.pvf_tracker_commit_many(self._h, ptr(handles(trks)), ptr(self._handles(frames)), n))

    def tracker_position(self, trk):
        b = np.zeros(4, np.float64)
        check(self._l.pvf_tracker_position(self._h, trk, ptr(b)))
        return tuple(b.tolist())

    def tracker_state(self, trk):
        F = np.zeros((32, 64, 64, 2), np.float64)
        A = np.zeros((32, 64, 64, 2), np.float64)
        B = np.zeros((64, 64), np.float64)
        check(self._l.pvf_debug_tracker_state(self._h, trk, ptr(F), ptr(A), ptr(B)))
        return F, A, B

    # ---- S4
    def landmarks(self, frames, boxes):
        n = len(boxes)
        pts = np.zeros((n, 68, 2), np.int32)
        if n == 0:
            return pts
        r = _rects(boxes, n)
        with self._staging():
            check(self._l.pvf_landmarks(self._h, ptr(self._handles(frames)), ptr(r), n, ptr(pts)))
        return pts

    def embed(self, frames, pts):
        pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 68, 2)
        n = len(pts)
        out = np.zeros((n, 128), np.float32)
        if n == 0:
            return out
        with self._staging():
            check(self._l.pvf_embed(self._h, ptr(self._handles(frames)), ptr(pts), n, ptr(out)))
        return out

    def landmarks_embed(self, frames, boxes):
        """landmarks() then embed() of the same faces in one library call: (int32 [n, 68, 2], float32 [n, 128])"""
        n = len(boxes)
        pts = np.zeros((n, 68, 2), np.int32)
        out = np.zeros((n, 128), np.float32)
        if n == 0:
            return pts, out
        r = _rects(boxes, n)
        with self._staging():
            check(self._l.pvf_landmarks_embed(self._h, ptr(self._handles(frames)), ptr(r), n, ptr(pts), ptr(out)))
        return pts, out

    def face_chips(self, frames, pts):
        pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 68, 2)
        n = len(pts)
        out = np.zeros((n, 150, 150, 3), np.uint8)
        with self._staging():
            check(self._l.pvf_face_chips(self._h, ptr(self._handles(frames)), ptr(pts), n, ptr(out)))
        return out

    def embed_chips(self, chips):
        chips = np.ascontiguousarray(chips, np.uint8).reshape(-1, 150, 150, 3)
        out = np.zeros((len(chips), 128), np.float32)
        check(self._l.pvf_embed_chips(self._h, ptr(chips), len(chips), ptr(out)))
        return out

    def extract_chip(self, frame, rect, cs, sn, rows, cols):
        f = self.stage(frame)
        r = np.asarray(rect, np.float64)
        out = np.zeros((rows, cols, 3), np.uint8)
        check(self._l.pvf_debug_extract_chip(self._h, f.handle, ptr(r), float(cs), float(sn), rows, cols, ptr(out)))
        return out

    # ---- f4: shot boundary detection
    def shot_dfd(self, frames, width, height, tables, want_gray=False, want_flow=False):
        """displaced frame differences of consecutive frames (structure/shot.py:71-99): float64 [n - 1]
        (+ the small gray images uint8 [n, height, width] and the flows float32 [n - 1, height, width, 2] on request)"""
        n = len(frames)
        t = np.ascontiguousarray(tables, np.float32)
        if t.shape != (22,):
            raise ValueError("shot tables: 22 floats (structure.shot_tables())")
        dfd = np.zeros(max(n - 1, 0), np.float64)
        gray = np.zeros((n, height, width), np.uint8) if want_gray else None
        flow = np.zeros((max(n - 1, 0), height, width, 2), np.float32) if want_flow else None
        with self._staging():
            check(self._l.pvf_shot_dfd(self._h, ptr(self._handles(frames)), n, int(width), int(height), ptr(t), ptr(dfd) if n > 1 else None,
                                       ptr(gray) if want_gray else None, ptr(flow) if want_flow and n > 1 else None))
        out = (dfd,)
        if want_gray:
            out += (gray,)
        if want_flow:
            out += (flow,)
        return out if len(out) > 1 else dfd

    # ---- S5
    def pair_mean_dist(self, X, row_start, metric=0):
        """T x T matrix of mean pair distances between the rows of two tracks; metric 0 = Euclidean (reference), 1 = cosine"""
        X = np.ascontiguousarray(X, np.float64)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        D = np.zeros((T, T), np.float64)
        check(self._l.pvf_pair_mean_dist_metric(self._h, ptr(X), X.shape[0], X.shape[1], ptr(rs), T, int(metric), ptr(D)))
        return D

    def cluster_tracks(self, X, row_start, threshold):
        X = np.ascontiguousarray(X, np.float64)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        labels = np.zeros(T, np.int32)
        log = np.zeros((max(T - 1, 1), 4), np.float64)
        n = C.c_int32(0)
        check(self._l.pvf_cluster_tracks(self._h, ptr(X), X.shape[0], X.shape[1], ptr(rs), T, float(threshold), ptr(labels), ptr(log), C.byref(n)))
        return labels, log[:n.value]

    @staticmethod
    def _f32_rows(emb):
        """(address, row stride in bytes, rows, on device?, keep-alive) of float32 descriptor rows: a numpy array or a DeviceRows"""
        if isinstance(emb, DeviceRows):
            return C.c_void_p(emb.ptr), emb.stride, emb.n, 1, emb
        a = np.asarray(emb)
        if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] != 128 or a.strides[1] != 4 or a.strides[0] < 512:
            a = np.ascontiguousarray(a, np.float32).reshape(-1, 128)
        return ptr(a), int(a.strides[0]) if len(a) else 512, len(a), 0, a

    def cluster_tracks_f32(self, emb, order, row_start, threshold, decimals=5, metric=0):
        """the in-memory clustering (pvf_cluster_tracks_f32): float32 descriptors (numpy [n, 128] or DeviceRows), rows of the table =
        round(emb[order], decimals) made on the device, upper-triangle pair means, mirror, agglomeration -> (labels, merge log)"""
        p, stride, n_src, on_dev, keep = self._f32_rows(emb)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        order = None if order is None else np.ascontiguousarray(order, np.int32)
        N = int(rs[-1])
        labels = np.zeros(T, np.int32)
        log = np.zeros((max(T - 1, 1), 4), np.float64)
        n = C.c_int32(0)
        check(self._l.pvf_cluster_tracks_f32(self._h, p, stride, n_src, on_dev, None if order is None else ptr(order), N, int(decimals), ptr(rs), T,
                                             int(metric), float(threshold), ptr(labels), ptr(log), C.byref(n)))
        return labels, log[:n.value]

    def pair_upper_rows_f32(self, emb, order, row_start, track0, track1, decimals=5, out=None):
        """the upper-triangle entries of rows [track0, track1) of the track-pair matrix, compact [(track1 - track0), T]: into `out`
        (a DeviceRows of T * 8-byte rows: stays in HBM for the all-gather) or returned as a numpy array"""
        p, stride, n_src, on_dev, keep = self._f32_rows(emb)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        order = None if order is None else np.ascontiguousarray(order, np.int32)
        m = int(track1) - int(track0)
        if out is None:
            res = np.zeros((max(m, 0), T), np.float64)
            op, odev = ptr(res), 0
        else:
            if out.n < m or out.stride != T * 8:
                raise ValueError("pair_upper_rows_f32: `out` must hold (track1 - track0) rows of T float64 values")
            res, op, odev = out, C.c_void_p(out.ptr), 1
        check(self._l.pvf_pair_upper_rows_f32(self._h, p, stride, n_src, on_dev, None if order is None else ptr(order), int(rs[-1]), int(decimals),
                                              ptr(rs), T, int(track0), int(track1), op, odev))
        return res

    def cluster_upper(self, U, row_start, threshold):
        """mirror + agglomeration of an assembled upper triangle: U numpy [T, T] or a DeviceRows of T rows of T float64"""
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        if isinstance(U, DeviceRows):
            if U.n != T or U.stride != T * 8:
                raise ValueError("cluster_upper: a T x T matrix is expected")
            p, on_dev = C.c_void_p(U.ptr), 1
        else:
            U = np.ascontiguousarray(U, np.float64)
            p, on_dev = ptr(U), 0
        labels = np.zeros(T, np.int32)
        log = np.zeros((max(T - 1, 1), 4), np.float64)
        n = C.c_int32(0)
        check(self._l.pvf_cluster_upper(self._h, p, on_dev, ptr(rs), T, float(threshold), ptr(labels), ptr(log), C.byref(n)))
        return labels, log[:n.value]

    def pair_mean_dist_rows(self, X, row_start, track0, track1):
        """the complete rows [track0, track1) of the T x T track-pair mean-distance matrix (other rows zero): j > i computed, j < i mirrored"""
        return self._pair_rows(self._l.pvf_pair_mean_dist_rows, X, row_start, track0, track1)

    def pair_upper_rows(self, X, row_start, track0, track1):
        """upper-triangle entries (j > i) of rows [track0, track1) of that matrix (everything else zero): a rank's share of a split clustering"""
        return self._pair_rows(self._l.pvf_pair_upper_rows, X, row_start, track0, track1)

    def _pair_rows(self, fn, X, row_start, track0, track1):
        X = np.ascontiguousarray(X, np.float64)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        D = np.zeros((T, T), np.float64)
        check(fn(self._h, ptr(X), X.shape[0], X.shape[1], ptr(rs), T, int(track0), int(track1), ptr(D)))
        return D

    def cluster_dist(self, D, row_start, threshold):
        D = np.ascontiguousarray(D, np.float64)
        rs = np.ascontiguousarray(row_start, np.int32)
        T = len(rs) - 1
        labels = np.zeros(T, np.int32)
        log = np.zeros((max(T - 1, 1), 4), np.float64)
        n = C.c_int32(0)
        check(self._l.pvf_cluster_dist(self._h, ptr(D), ptr(rs), T, float(threshold), ptr(labels), ptr(log), C.byref(n)))
        return labels, log[:n.value]

    # ---- measurement
    def detector_screening(self, on=True, list_cap=0):
        """the detector's screening pass (csrc/screen.hip): windows are first scored on the f16 matrix cores with a proven error bound and
        only those within the bound of the threshold go through the exact fp32 chain -- same rectangles, order and scores, bit for bit"""
        check(self._l.pvf_detector_screening(self._h, 1 if on else 0, int(list_cap)))

    def detector_screening_stats(self):
        """{"batches", "listed", "retries", "bounds", "pipe_err"}: batches screened, (window, filter) pairs that went through the exact
        chain, calls repeated on the dense kernel, the error bound per filter (score units), the accumulation error the context measured
        on its matrix pipe (relative to the sum of magnitudes; -1 before the first screened batch)"""
        b, n, r, pe = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_double(-1.0)
        bounds = np.zeros(8, np.float64)
        check(self._l.pvf_detector_screening_stats(self._h, C.byref(b), C.byref(n), C.byref(r), ptr(bounds), C.byref(pe)))
        return {"batches": b.value, "listed": n.value, "retries": r.value, "bounds": bounds[:5].tolist(), "pipe_err": pe.value}

    def prof_enable(self, on=True):
        check(self._l.pvf_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        check(self._l.pvf_prof_reset(self._h))

    def prof_get(self, family):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._l.pvf_prof_get(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


_default = None


def default_context():
    """Process-wide context on the GPU of this rank (LOCAL_RANK) -- what the dlib-like shim objects use."""
    global _default
    if _default is None:
        import os
        _default = Context(device=int(os.environ.get("LOCAL_RANK", "0")))
    return _default
