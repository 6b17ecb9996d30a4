"""ctypes binding of libpvface.so (include/pvface.h).  No CPU fallback: if the library is missing the import of any
compute entry point fails loudly, and `Context()` fails when no gfx950 device is visible."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PVF_LIBRARY: another build of the SAME library (the address-sanitizer build of `make asan`, tests/README); never a different implementation
LIB_PATH = os.environ.get("PVF_LIBRARY") or os.path.join(_HERE, "libpvface.so")

_lib = None


class PvfError(RuntimeError):
    pass


class Rect(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("right", C.c_int32), ("bottom", C.c_int32)]


H = C.c_uint64
P = C.c_void_p
_SIGS = {
    "pvf_last_error": (C.c_char_p, []),
    "pvf_version": (C.c_int32, []),
    "pvf_device_count": (C.c_int32, [P]),
    "pvf_ctx_create": (C.c_int32, [C.c_int32, P]),
    "pvf_ctx_create_prio": (C.c_int32, [C.c_int32, C.c_int32, P]),
    "pvf_ctx_destroy": (C.c_int32, [H]),
    "pvf_sync": (C.c_int32, [H]),
    "pvf_load_detector": (C.c_int32, [H, C.c_char_p]),
    "pvf_load_shape_predictor": (C.c_int32, [H, C.c_char_p]),
    "pvf_load_embedder": (C.c_int32, [H, C.c_char_p]),
    "pvf_model_tensor": (C.c_int32, [C.c_char_p, C.c_int32, C.c_char_p, P, C.c_int64, P]),
    "pvf_set_tracker_tables": (C.c_int32, [H, P, P, P, P, C.c_double, C.c_double]),
    "pvf_frame_upload": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_int64, P]),
    "pvf_frame_wrap_device": (C.c_int32, [H, P, C.c_int32, C.c_int32, P]),
    "pvf_frame_release": (C.c_int32, [H, H]),
    "pvf_frame_release_many": (C.c_int32, [H, P, C.c_int32]),
    "pvf_frame_pool_trim": (C.c_int32, [H, C.c_int64, P]),
    "pvf_mem_info": (C.c_int32, [H, P, P]),
    "pvf_frame_device_ptr": (C.c_int32, [H, H, P]),
    "pvf_ingest_create": (C.c_int32, [H, C.c_int32, C.c_int32, C.c_int32, P]),
    "pvf_ingest_destroy": (C.c_int32, [H, H]),
    "pvf_ingest_acquire": (C.c_int32, [H, H, P, P]),
    "pvf_ingest_submit": (C.c_int32, [H, H, C.c_int32, P]),
    "pvf_ingest_wait": (C.c_int32, [H, H]),
    "pvf_frame_resize": (C.c_int32, [H, H, C.c_int32, C.c_int32, P]),
    "pvf_detect": (C.c_int32, [H, H, C.c_int32, C.c_double, P, P, C.c_int32, P]),
    "pvf_detect_batch": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_double, P, P, P, C.c_int32]),
    "pvf_detect_many": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_int32, C.c_double, P, P, P, C.c_int32]),
    "pvf_detector_screening": (C.c_int32, [H, C.c_int32, C.c_int32]),
    "pvf_detector_screening_stats": (C.c_int32, [H, P, P, P, P, P]),
    "pvf_tracker_create": (C.c_int32, [H, P]),
    "pvf_tracker_destroy": (C.c_int32, [H, H]),
    "pvf_tracker_create_many": (C.c_int32, [H, C.c_int32, P]),
    "pvf_tracker_destroy_many": (C.c_int32, [H, P, C.c_int32]),
    "pvf_tracker_clone_many": (C.c_int32, [H, P, C.c_int32, P]),
    "pvf_tracker_start": (C.c_int32, [H, H, H, P]),
    "pvf_tracker_update": (C.c_int32, [H, H, H, P]),
    "pvf_tracker_position": (C.c_int32, [H, H, P]),
    "pvf_tracker_start_many": (C.c_int32, [H, P, P, P, C.c_int32]),
    "pvf_tracker_update_many": (C.c_int32, [H, P, P, C.c_int32, P, P]),
    "pvf_tracker_update_many_deferred": (C.c_int32, [H, P, P, C.c_int32, P, P]),
    "pvf_tracker_commit_many": (C.c_int32, [H, P, P, C.c_int32]),
    "pvf_overlap_matrix": (C.c_int32, [P, C.c_int32, P, C.c_int32, C.c_double, P]),
    "pvf_munkres": (C.c_int32, [P, C.c_int32, P]),
    "pvf_associate": (C.c_int32, [P, C.c_int32, P, C.c_int32, C.c_double, P]),
    "pvf_landmarks": (C.c_int32, [H, P, P, C.c_int32, P]),
    "pvf_embed": (C.c_int32, [H, P, P, C.c_int32, P]),
    "pvf_landmarks_embed": (C.c_int32, [H, P, P, C.c_int32, P, P]),
    "pvf_embed_chips": (C.c_int32, [H, P, C.c_int32, P]),
    "pvf_face_chips": (C.c_int32, [H, P, P, C.c_int32, P]),
    "pvf_pair_mean_dist": (C.c_int32, [H, P, C.c_int32, C.c_int32, P, C.c_int32, P]),
    "pvf_pair_mean_dist_metric": (C.c_int32, [H, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, P]),
    "pvf_pair_mean_dist_rows": (C.c_int32, [H, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, C.c_int32, P]),
    "pvf_pair_upper_rows": (C.c_int32, [H, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, C.c_int32, P]),
    "pvf_lane_create": (C.c_int32, [C.c_int32, P, P, C.c_int32, C.c_double, C.c_double, C.c_int32, P]),
    "pvf_lane_destroy": (C.c_int32, [H]),
    "pvf_lane_feed_plan": (C.c_int32, [H, C.c_int32, C.c_int32, P, P, P, P]),
    "pvf_lane_advance": (C.c_int32, [H, P, P, C.c_int32, P, P, P, C.c_int32, P, P]),
    "pvf_lane_take_dead": (C.c_int32, [H, P, C.c_int32, P]),
    "pvf_lane_edges": (C.c_int32, [H, P, P, P, P, P, P, C.c_int32]),
    "pvf_shot_tracks": (C.c_int32, [H, H, P, C.c_int32, C.c_double, P, C.c_int32, P, P, C.c_int32, P]),
    "pvf_track_rows": (C.c_int32, [P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "pvf_round_decimals": (C.c_int32, [P, C.c_int64, C.c_int32, P]),
    "pvf_cluster_dist": (C.c_int32, [H, P, P, C.c_int32, C.c_double, P, P, P]),
    "pvf_cluster_tracks": (C.c_int32, [H, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_double, P, P, P]),
    "pvf_cluster_upper": (C.c_int32, [H, P, C.c_int32, P, C.c_int32, C.c_double, P, P, P]),
    "pvf_cluster_tracks_f32": (C.c_int32, [H, P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, C.c_double, P, P, P]),
    "pvf_pair_upper_rows_f32": (C.c_int32, [H, P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, C.c_int32, P, C.c_int32]),
    "pvf_format_rows": (C.c_int32, [P, P, P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int64, P]),
    "pvf_round_rows": (C.c_int32, [P, C.c_int64, C.c_int32, P]),
    "pvf_parse_rows": (C.c_int32, [C.c_char_p, C.c_int64, P, C.c_int64, P, P]),
    "pvf_prof_enable": (C.c_int32, [H, C.c_int32]),
    "pvf_prof_reset": (C.c_int32, [H]),
    "pvf_prof_get": (C.c_int32, [H, C.c_char_p, P, P]),
    "pvf_debug_pyramid_level": (C.c_int32, [H, H, C.c_int32, C.c_int32, P, P, P]),
    "pvf_debug_pyramid_batch": (C.c_int32, [H, P, C.c_int32, C.c_int32]),
    "pvf_debug_level_features": (C.c_int32, [H, H, C.c_int32, C.c_int32, P, P, P]),
    "pvf_debug_fhog": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P, P]),
    "pvf_debug_detect_raw": (C.c_int32, [H, H, C.c_int32, C.c_double, P, P, C.c_int32, P]),
    "pvf_debug_detect_raw_many": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_int32, C.c_double, P, P, C.c_int64, P]),
    "pvf_debug_extract_chip": (C.c_int32, [H, H, P, C.c_double, C.c_double, C.c_int32, C.c_int32, P]),
    "pvf_debug_tracker_state": (C.c_int32, [H, H, P, P, P]),
    "pvf_shot_dfd": (C.c_int32, [H, P, C.c_int32, C.c_int32, C.c_int32, P, P, P, P]),
}
EXPORTS = sorted(_SIGS)


def lib():
    """Load libpvface.so (built by `make -C pyannote-video_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PvfError("libpvface.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)   # AttributeError if an export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise PvfError(lib().pvf_last_error().decode("utf-8", "replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def handles(seq):
    return np.ascontiguousarray(np.asarray(seq, dtype=np.uint64))


def device_count():
    n = C.c_int32(0)
    check(lib().pvf_device_count(C.byref(n)))
    return n.value


# ---- host-only helpers (usable without a GPU) ---------------------------------------------------------
def model_tensor(path, kind, name, dtype):
    """one tensor of a model file as the C loaders parse it (kind 1 = shape predictor, 2 = embedder)"""
    n = C.c_int64(0)
    check(lib().pvf_model_tensor(str(path).encode(), int(kind), name.encode(), None, 0, C.byref(n)))
    out = np.zeros(n.value // np.dtype(dtype).itemsize, dtype)
    check(lib().pvf_model_tensor(str(path).encode(), int(kind), name.encode(), ptr(out), out.nbytes, C.byref(n)))
    return out


def overlap_matrix(a, b, ratio):
    a = np.ascontiguousarray(a, np.float64).reshape(-1, 4)
    b = np.ascontiguousarray(b, np.float64).reshape(-1, 4)
    out = np.zeros((len(a), len(b)), np.float64)
    check(lib().pvf_overlap_matrix(ptr(a), len(a), ptr(b), len(b), float(ratio), ptr(out)))
    return out


import threading as _threading
_assoc_tls = _threading.local()       # ctypes releases the GIL inside pvf_associate: the reused buffers are per thread
_ASSOC_KEEP = 256


def associate(trackers, detections, ratio):
    """[(tracker row, detection index)] of the reference's _associate, tracker rows ascending.
    Called once per frame and pass by the tracking state machine (thousands of times per shot): plain ctypes arrays, reused per size
    (per thread, at most _ASSOC_KEEP sizes)."""
    na, nb = len(trackers), len(detections)
    key = (na, nb)
    cache = getattr(_assoc_tls, "buf", None)
    if cache is None:
        cache = _assoc_tls.buf = {}
    buf = cache.get(key)
    if buf is None:
        if len(cache) >= _ASSOC_KEEP:
            cache.clear()
        buf = cache[key] = ((C.c_double * (4 * na))(), (C.c_double * (4 * nb))(), (C.c_int32 * max(na, 1))())
    a, b, out = buf
    k = 0
    for box in trackers:
        a[k], a[k + 1], a[k + 2], a[k + 3] = box
        k += 4
    k = 0
    for box in detections:
        b[k], b[k + 1], b[k + 2], b[k + 3] = box
        k += 4
    check(lib().pvf_associate(a, na, b, nb, float(ratio), out))
    return [(t, d) for t, d in enumerate(out[:na]) if d >= 0]


def round_decimals(x, decimals):
    """round(float(v), decimals) for every v of an array (Python's round: the correctly rounded decimal, ties decided on the exact binary
    value) -- pvf_round_decimals"""
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    if x.size:
        check(lib().pvf_round_decimals(ptr(x), x.size, int(decimals), ptr(out)))
    return out


def track_rows(boxes, det_width, det_height, width, height):
    """integer boxes [n, 4] of track rows -> (file_box float64 [n, 4] = float32('%.3f' % (box / detection size)), pixel_box int32 [n, 4] =
    int(file_box * frame size)): what `pyannote-face track` writes and `extract` rebuilds (pvf_track_rows)"""
    b = np.ascontiguousarray(boxes, np.int32).reshape(-1, 4)
    fb = np.empty(b.shape, np.float64)
    pb = np.empty(b.shape, np.int32)
    if len(b):
        check(lib().pvf_track_rows(ptr(b), len(b), int(det_width), int(det_height), int(width), int(height), ptr(fb), ptr(pb)))
    return fb, pb


def format_rows(t, identifier, values, decimals=5):
    """bytes of the lines "{t:.3f} {identifier:d}" + " {v:.<decimals>f}" per value (landmarks.txt / embedding.txt rows)"""
    t = np.ascontiguousarray(t, np.float64)
    ident = np.ascontiguousarray(identifier, np.int64)
    v = np.ascontiguousarray(values, np.float64)
    v = v.reshape(len(t), -1) if len(t) else v.reshape(0, max(v.shape[-1], 1) if v.ndim else 1)
    cap = len(t) * (64 + 42 * v.shape[1])
    buf = np.empty(max(cap, 1), np.uint8)
    n = C.c_int64(0)
    check(lib().pvf_format_rows(ptr(t), ptr(ident), ptr(v), len(t), v.shape[1], int(decimals), ptr(buf), cap, C.byref(n)))
    return buf[:n.value].tobytes()


def round_rows(x, decimals=5):
    """np.round(x.astype(float64), decimals) for a float32 array, by the library (pvf_round_rows)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.float64)
    check(lib().pvf_round_rows(ptr(x), x.size, int(decimals), ptr(out)))
    return out


def parse_rows(text):
    """float64 [rows, cols] of whitespace-separated numeric text (bytes): np.loadtxt(..., ndmin=2) for the files format_rows writes"""
    n = len(text)
    rows, cols = C.c_int64(0), C.c_int32(0)
    for cap in (n // 6 + 16, n // 2 + 1):             # (a value and its separator take >= 2 bytes; the files here use >= 8)
        out = np.empty(cap, np.float64)
        if lib().pvf_parse_rows(text, n, ptr(out), out.size, C.byref(rows), C.byref(cols)) == 0:
            return out[:rows.value * cols.value].reshape(rows.value, cols.value)
        if b"too small" not in lib().pvf_last_error():
            break
    raise PvfError(lib().pvf_last_error().decode("utf-8", "replace"))


def munkres(cost):
    """Munkres().compute(cost) for a square matrix: list of (row, column)"""
    cost = np.ascontiguousarray(cost, np.float64)
    n = cost.shape[0]
    assert cost.shape == (n, n)
    out = np.zeros(n, np.int32)
    check(lib().pvf_munkres(ptr(cost), n, ptr(out)))
    return [(i, int(out[i])) for i in range(n)]
