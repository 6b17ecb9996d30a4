"""Forward/backward tracking by detection -- host state machine of the reference, re-stated around batched GPU calls.

Reference: pyannote/video/tracking.py (TrackingByDetection :68-434).  Semantics kept verbatim because they decide the
integer outputs (track ids, boxes, status strings):
  - shot membership / flush order                      (:44-58, :406-417)
  - graph nodes (t, (l,t,r,b), status), same insertion order of nodes and edges   (:214-259, :420-427)
  - association = overlap gating on BOTH boxes + Munkres on max-overlap cost      (:129-182)
  - _fix: per-timestamp mean + banker's int(round()), status join order, error()  (:261-296)
  - _fill_gaps on (last, first) items in (min_t, max_t) order                     (:298-329)
  - tracks yielded sorted by (min_t, max_t), normalised by the frame size          (:331-372)

What is different is *how* the work is issued: detections of a whole shot are computed in batches before the passes,
and the forward and backward passes (of one shot, or of many shots) advance in lock-step so that every frame step issues
ONE batched `tracker.update` and ONE batched `start_track` for all lanes (pvf_tracker_update_many / _start_many).
The passes only communicate through the graph, and each lane's graph mutations are replayed in the reference's order
(forward pass first, then backward), so the resulting graph is identical.
"""
from __future__ import division
import contextlib
import itertools
import numpy as np
import networkx as nx
from . import _lib
from .shim import drectangle

FORWARD = 'forward'
BACKWARD = 'backward'
DETECTION = 'detection'
ERROR = 'error'

_STATUS_ORDER = {FORWARD: 1, DETECTION: 2, BACKWARD: 3}


def get_segment_generator(segmentation):
    """Time-driven segment generator (tracking.py:44-58): send(t) returns the end of a segment once t has passed it.
    Segments need `.end`; (start, end) tuples are accepted too."""
    t = yield
    for segment in segmentation:
        T = segment.end if hasattr(segment, 'end') else segment[1]
        while True:
            if T > t:
                t = yield
                continue
            t = yield T
            break


def get_min_max_t(track):
    return (min(t for t, _, _ in track), max(t for t, _, _ in track))


def _floats(a):
    """confidences of a batch as Python floats (the values float(a[k]) gives, in one conversion)"""
    return a.tolist() if hasattr(a, "tolist") else [float(v) for v in a]


def _rows(a):
    """positions of a batch as tuples of Python floats (tuple(float(v) for v in a[k]) for every row, in one conversion)"""
    return [tuple(r) for r in a.tolist()] if hasattr(a, "tolist") else [tuple(float(v) for v in r) for r in a]


class HipTrackers(object):
    """Batched tracker backend on one Context (default)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def start_many(self, frames, boxes):
        hs = self.ctx.tracker_create_many(len(boxes))
        self.ctx.tracker_start_many(hs, frames, boxes)
        return hs

    def update_many(self, handles, frames, defer=False):
        return self.ctx.tracker_update_many(handles, frames, defer)

    def commit_many(self, handles, frames):
        """filter update of trackers whose last (deferred) update ran on `frames`"""
        self.ctx.tracker_commit_many(handles, frames)

    def release(self, handle):
        self.ctx.tracker_destroy(handle)

    def release_many(self, handles):
        self.ctx.tracker_destroy_many(handles)

    def speculate(self, cache, detections_at, chunk=4096):
        """Issue EVERY start_track of a lane and the first update of every started tracker as large batches.

        In the reference loop (tracking.py:199-259) a tracker started on the detections of frame i is unconditionally
        updated on frame i+1 (:202-203) before anything can kill it, so these two operations never depend on the
        association logic; each tracker's arithmetic is independent of the batch it runs in, so the results are the
        ones the frame-by-frame order produces.  Returns {t: (handles, psr or None, boxes or None)}.

        Backends with `commit_many` run these first updates WITHOUT the filter update (a tracker that is matched to a
        detection on that frame is dropped, tracking.py:219-224, which is the common case); the lane commits it for the
        trackers that live on, before their next update."""
        defer = hasattr(self, 'commit_many')
        flat_f, flat_b, owner = [], [], []
        for i, (t, frame) in enumerate(cache):
            for d in detections_at.get(t, []):
                flat_f.append(frame)
                flat_b.append(tuple(float(v) for v in d))
                owner.append(i)
        n = len(flat_b)
        hs = []
        for o in range(0, n, chunk):
            hs.extend(self.start_many(flat_f[o:o + chunk], flat_b[o:o + chunk]))
        last = len(cache) - 1
        upd = [k for k in range(n) if owner[k] < last]
        psr = np.zeros(n, np.float64)
        pos = np.zeros((n, 4), np.float64)
        for o in range(0, len(upd), chunk):
            ks = upd[o:o + chunk]
            if defer:
                p, b = self.update_many([hs[k] for k in ks], [cache[owner[k] + 1][1] for k in ks], True)
            else:
                p, b = self.update_many([hs[k] for k in ks], [cache[owner[k] + 1][1] for k in ks])
            psr[ks] = p
            pos[ks] = b
        plan, k = {}, 0
        for i, (t, _) in enumerate(cache):
            m = len(detections_at.get(t, []))
            if m:
                plan[t] = (hs[k:k + m], psr[k:k + m] if i < last else None, pos[k:k + m] if i < last else None)
                k += m
        return plan


    def speculate_pair(self, cache, detections_at, chunk=4096, counts=None, boxes=None):
        """speculate() for the forward AND the backward pass of a shot: (plan_forward, plan_backward).

        Both passes start one tracker per detection from the same frame and box, i.e. with bit-identical filters, so the
        trackers are started once and cloned for the second pass (clones share the filters until one side writes them: no device
        work); only the first updates differ (frame i+1 for the forward pass, frame i-1 for the backward one).
        Frame handles, owners and boxes are arrays built once per shot; every batched call indexes them.  A caller that already holds
        the detections as arrays passes `counts` (per frame of the cache) and `boxes` ([sum(counts), 4], frame by frame) instead of the dict."""
        ctx = self.ctx
        if counts is None:
            counts = [len(detections_at.get(t, ())) for t, _ in cache]
            boxes = np.array([d for t, _ in cache for d in detections_at.get(t, ())], np.float64).reshape(-1, 4)
        else:
            counts = [int(v) for v in counts]
            boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
        # numpy frames are staged on first sight and cached by identity; the handles looked up here are used by every batched call
        # below, so nothing staged may be evicted until the last of them has run (a shot longer than the cache's capacity)
        hold = ctx._staging() if hasattr(ctx, "_staging") else contextlib.nullcontext()
        with hold:
            return self._speculate_pair(ctx, cache, counts, boxes, chunk)

    def _speculate_pair(self, ctx, cache, counts, boxes, chunk):
        n = sum(counts)
        fh = ctx.frame_handles([f for _, f in cache])
        owner = np.repeat(np.arange(len(cache)), counts)
        hs_f = ctx.tracker_create_many(n, as_array=True)
        for o in range(0, n, chunk):
            ctx.tracker_start_many(hs_f[o:o + chunk], fh[owner[o:o + chunk]], boxes[o:o + chunk])
        hs_b = ctx.tracker_clone_many(hs_f, as_array=True)          # before any update touches the originals
        last = len(cache) - 1
        starts = np.concatenate([[0], np.cumsum(counts)])
        # the first updates of both passes in ONE batched call per chunk (forward trackers on frame i + 1, their clones on frame i - 1):
        # one set of launches and one wait for results instead of two
        sides = ((hs_f, 1, last), (hs_b, -1, 0))
        upd = [np.nonzero(owner != edge)[0] for _, _, edge in sides]
        all_h = np.concatenate([hs[u] for (hs, _, _), u in zip(sides, upd)]) if n else np.zeros(0, np.uint64)
        all_f = np.concatenate([fh[owner[u] + step] for (_, step, _), u in zip(sides, upd)]) if n else np.zeros(0, np.uint64)
        all_p = np.zeros(len(all_h), np.float64)
        all_b = np.zeros((len(all_h), 4), np.float64)
        for o in range(0, len(all_h), chunk):
            p, b = ctx.tracker_update_many(all_h[o:o + chunk], all_f[o:o + chunk], True)
            all_p[o:o + chunk] = p
            all_b[o:o + chunk] = b
        plans, o = [], 0
        times = [t for t, _ in cache]
        for (hs, step, edge), u in zip(sides, upd):
            psr = np.zeros(n, np.float64)
            pos = np.zeros((n, 4), np.float64)
            psr[u] = all_p[o:o + len(u)]
            pos[u] = all_b[o:o + len(u)]
            o += len(u)
            plans.append(ArrayPlan(times, counts, starts, hs, psr, pos, edge))
        return plans[0], plans[1]

    def speculate_window(self, fh, owner, boxes, n_frames, chunk=4096):
        """One window of ONE pass (engine.WindowedPlan): trackers for the detections `boxes` (row k lives on the frame with processing
        index owner[k]; `fh` = the shot's frame handles in processing order), started and -- unless their frame is the pass's last --
        updated once on the next frame, filter update deferred.  Returns (handles, psr, positions) as arrays.  A long shot (or a crowded
        one) asks for its trackers window by window, so that the 2.39 MB of filters per detection exist for a window, not for the shot."""
        ctx = self.ctx
        n = len(owner)
        boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
        hs = ctx.tracker_create_many(n, as_array=True)
        for o in range(0, n, chunk):
            ctx.tracker_start_many(hs[o:o + chunk], fh[owner[o:o + chunk]], boxes[o:o + chunk])
        psr = np.zeros(n, np.float64)
        pos = np.zeros((n, 4), np.float64)
        upd = np.nonzero(owner != n_frames - 1)[0]
        for o in range(0, len(upd), chunk):
            ks = upd[o:o + chunk]
            p, b = ctx.tracker_update_many(hs[ks], fh[owner[ks] + 1], True)
            psr[ks] = p
            pos[ks] = b
        return hs, psr, pos


class ArrayPlan(object):
    """plan[t] -> (handles, psr or None, positions or None) of the trackers started on the detections of the frame at time t, read out
    of the arrays a bulk call filled.  The per-frame views are made when the lane asks for them, i.e. in the tracking thread: the
    thread that feeds the GPU hands the arrays over and goes on to the next batch."""

    def __init__(self, times, counts, starts, handles, psr, pos, edge):
        self.times, self.counts, self.starts, self.hs, self.psr, self.pos, self.edge = times, counts, starts, handles, psr, pos, edge
        self._index = self._hl = None

    def native_feed(self, lane, p):
        """the whole plan in the lane's processing order (forward: as stored; backward: the frames reversed, a frame's trackers in place)"""
        n = len(self.counts)
        if lane.direction == BACKWARD:
            starts = self.starts
            idx = np.concatenate([np.arange(starts[i], starts[i + 1]) for i in range(n - 1, -1, -1)]) if len(self.hs) else np.zeros(0, np.int64)
            has = np.arange(n - 1, -1, -1) != self.edge
            lane.feed(0, has, self.hs[idx], self.psr[idx], self.pos[idx])
        else:
            lane.feed(0, np.arange(n) != self.edge, self.hs, self.psr, self.pos)

    def __getitem__(self, t):
        if self._index is None:
            self._index = {t: i for i, t in enumerate(self.times)}
            self._hl = self.hs.tolist()
        i = self._index[t]
        m = self.counts[i]
        if not m:
            raise KeyError(t)
        k = int(self.starts[i])
        if i == self.edge:
            return self._hl[k:k + m], None, None
        return self._hl[k:k + m], self.psr[k:k + m], self.pos[k:k + m]


_KIND = {FORWARD: 1, DETECTION: 2, BACKWARD: 3}
_KIND_NAME = {1: FORWARD, 2: DETECTION, 3: BACKWARD}
_status_cache = {}


def status_of(code):
    """status string of a track row from the library's code (csrc/shotgraph.hip): counts of forward / detection / backward nodes of the
    timestamp joined in the reference's order (tracking.py:282-287), wrapped in error(...) when two of its boxes do not overlap"""
    s = _status_cache.get(code)
    if s is None:
        parts = [FORWARD] * (code & 255) + [DETECTION] * ((code >> 8) & 255) + [BACKWARD] * ((code >> 16) & 255)
        s = "+".join(parts)
        if code >> 24:
            s = "error({0})".format(s)
        _status_cache[code] = s
    return s


def detection_arrays(dets):
    """[[box]] per frame -> (counts int32 [n], boxes float64 [sum, 4]): what the library's lanes take"""
    counts = np.array([len(d) for d in dets], np.int32)
    boxes = np.array([b for d in dets for b in d], np.float64).reshape(-1, 4)
    return counts, boxes


class NativeLane(object):
    """one pass over one shot run by the library (csrc/shotgraph.hip: the state machine of `_lane`, array in / array out); this object
    feeds it the plan and serves its requests through the lane protocol of `_lane`, so the same scheduler drives both forms"""

    def __init__(self, n_frames, counts, boxes, direction, min_confidence, ratio, deferring=True):
        import ctypes as C
        self.n = int(n_frames)
        self.counts = np.ascontiguousarray(counts, np.int32)
        boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 4)
        self.total = int(self.counts.sum())
        self.boxes = boxes
        self.direction = direction
        h = C.c_uint64(0)
        _lib.check(_lib.lib().pvf_lane_create(self.n, _lib.ptr(self.counts), _lib.ptr(boxes), _KIND[direction], float(min_confidence), float(ratio),
                                              1 if deferring else 0, C.byref(h)))
        self.handle = h.value
        cap = max(self.total, 1)
        self._req_h = np.zeros(cap, np.uint64)
        self._req_f = np.zeros(cap, np.int32)
        self._dead = np.zeros(max(self.total, 1), np.uint64)

    def close(self):
        if self.handle:
            _lib.lib().pvf_lane_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def feed(self, p0, has_update, handles, psr, pos):
        has_update = np.ascontiguousarray(has_update, np.uint8)
        handles = np.ascontiguousarray(handles, np.uint64)
        psr = np.ascontiguousarray(psr, np.float64)
        pos = np.ascontiguousarray(pos, np.float64)
        _lib.check(_lib.lib().pvf_lane_feed_plan(self.handle, int(p0), len(has_update), _lib.ptr(has_update), _lib.ptr(handles), _lib.ptr(psr), _lib.ptr(pos)))

    def advance(self, reply=None):
        """-> (request, handles uint64 [k], shot frame indices int32 [k], plan_from)"""
        import ctypes as C
        req, n_req, pf = C.c_int32(0), C.c_int32(0), C.c_int32(-1)
        if reply is None:
            rp, rb, nr = None, None, 0
        else:
            rp = np.ascontiguousarray(reply[0], np.float64)
            rb = np.ascontiguousarray(reply[1], np.float64).reshape(-1, 4)
            nr = len(rp)
        _lib.check(_lib.lib().pvf_lane_advance(self.handle, _lib.ptr(rp) if nr else None, _lib.ptr(rb) if nr else None, nr, C.byref(req),
                                               _lib.ptr(self._req_h), _lib.ptr(self._req_f), len(self._req_h), C.byref(n_req), C.byref(pf)))
        k = n_req.value
        return req.value, self._req_h[:k].copy(), self._req_f[:k].copy(), pf.value

    def take_dead(self):
        import ctypes as C
        n = C.c_int32(0)
        _lib.check(_lib.lib().pvf_lane_take_dead(self.handle, _lib.ptr(self._dead), len(self._dead), C.byref(n)))
        return self._dead[:n.value].tolist()

    def edges(self, times):
        """the pass's add_edge calls as the Python form records them: [(u, v, confidence)] with nodes (t, box, status)"""
        import ctypes as C
        n = C.c_int32(0)
        _lib.check(_lib.lib().pvf_lane_edges(self.handle, C.byref(n), None, None, None, None, None, 0))
        k = n.value
        ufk, vfk = np.zeros((k, 2), np.int32), np.zeros((k, 2), np.int32)
        ub, vb, cf = np.zeros((k, 4), np.float64), np.zeros((k, 4), np.float64), np.zeros(k, np.float64)
        if k:
            _lib.check(_lib.lib().pvf_lane_edges(self.handle, C.byref(n), _lib.ptr(ufk), _lib.ptr(ub), _lib.ptr(vfk), _lib.ptr(vb), _lib.ptr(cf), k))

        def node(fk, b):
            box = tuple(int(v) for v in b) if fk[1] == 2 else tuple(b)
            return (times[fk[0]], box, _KIND_NAME[fk[1]])
        return [(node(ufk[i].tolist(), ub[i].tolist()), node(vfk[i].tolist(), vb[i].tolist()), float(cf[i])) for i in range(k)]


def shot_tracks_native(lane_forward, lane_backward, times, max_gap):
    """tracks of a shot from its two finished native passes: (rows int32 [m, 6] = frame, l, t, r, b, status code; track_start int32 [T + 1])"""
    import ctypes as C
    times = np.ascontiguousarray(times, np.float64)
    # room known beforehand: a row is a graph node, a node is a detection or the far end of an edge of one of the two passes (with
    # detections every N frames or long tracker bridges a shot has many times more rows than detections: sized from the detections
    # alone the whole union-find / _fix / _fill_gaps pass ran twice -- ADVICE r5).  pvf_lane_edges with no room reports the count.
    n_f, n_b = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.lib().pvf_lane_edges(lane_forward.handle, C.byref(n_f), None, None, None, None, None, 0))
    _lib.check(_lib.lib().pvf_lane_edges(lane_backward.handle, C.byref(n_b), None, None, None, None, None, 0))
    cap = max(lane_forward.total, 1) + n_f.value + n_b.value + 16
    while True:
        rows = np.zeros((cap, 6), np.int32)
        starts = np.zeros(cap + 1, np.int32)
        n_rows, n_tracks = C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.lib().pvf_shot_tracks(lane_forward.handle, lane_backward.handle, _lib.ptr(times), len(times), float(max_gap), _lib.ptr(rows), cap,
                                              C.byref(n_rows), _lib.ptr(starts), cap, C.byref(n_tracks)))
        if n_rows.value <= cap and n_tracks.value <= cap:
            return rows[:n_rows.value], starts[:n_tracks.value + 1]
        cap = max(n_rows.value, n_tracks.value) + 16


class ObjectTrackers(object):
    """Adapter for any per-object tracker class with dlib's start_track / update / get_position (test seam, S2)."""

    def __init__(self, factory, rect_factory=drectangle):
        self.factory, self.rect = factory, rect_factory

    def start_many(self, frames, boxes):
        out = []
        for f, b in zip(frames, boxes):
            t = self.factory()
            t.start_track(f, self.rect(*b))
            out.append(t)
        return out

    def update_many(self, handles, frames):
        psr = np.array([h.update(f) for h, f in zip(handles, frames)], np.float64)
        pos = []
        for h in handles:
            p = h.get_position()
            pos.append((p.left(), p.top(), p.right(), p.bottom()))
        return psr, np.array(pos, np.float64).reshape(-1, 4)

    def release(self, handle):
        pass


class LaneScheduler(object):
    """Runs any number of lane coroutines in lock-step; lanes may be added between rounds (shots whose detections
    became available).  One round = every lane's pending request served with ONE update batch and ONE start batch."""

    def __init__(self, backend):
        self.backend = backend
        self.lanes = {}
        self.pending = {}
        self._next = 0

    def __len__(self):
        return len(self.pending)

    def add(self, gen):
        k = self._next
        self._next += 1
        try:
            self.pending[k] = next(gen)
            self.lanes[k] = gen
        except StopIteration:
            pass
        return k

    def round(self):
        pending, backend = self.pending, self.backend
        replies = {}
        upd = [(k, r) for k, r in pending.items() if r[0] == 'update']
        if upd:
            hs = [h for _, r in upd for h in r[1]]
            fr = [f for _, r in upd for f in r[2]]
            psr, boxes = backend.update_many(hs, fr)
            o = 0
            for k, r in upd:
                n = len(r[1])
                replies[k] = (psr[o:o + n], boxes[o:o + n])
                o += n
        cm = [(k, r) for k, r in pending.items() if r[0] == 'commit']
        if cm:
            backend.commit_many([h for _, r in cm for h in r[1]], [f for _, r in cm for f in r[2]])
            for k, _ in cm:
                replies[k] = None
        st = [(k, r) for k, r in pending.items() if r[0] == 'start']
        if st:
            fr = [f for _, r in st for f in r[1]]
            bx = [b for _, r in st for b in r[2]]
            hs = backend.start_many(fr, bx)
            o = 0
            for k, r in st:
                n = len(r[2])
                replies[k] = hs[o:o + n]
                o += n
        for k, r in pending.items():
            if r[0] == 'release':
                backend.release(r[1])
                replies[k] = None
        nxt = {}
        for k in pending:
            try:
                nxt[k] = self.lanes[k].send(replies[k])
            except StopIteration:
                del self.lanes[k]
        self.pending = nxt


class TrackingByDetection(object):
    """(Forward/backward) tracking by detection -- same constructor and call contract as the reference class."""

    def __init__(self, detect_func, detect_smallest=1, detect_min_size=0., detect_every=0.,
                 track_min_confidence=10., track_min_overlap_ratio=0.3, track_max_gap=0.,
                 trackers=None, detect_batch_func=None, detect_batch_size=8):
        super(TrackingByDetection, self).__init__()
        self.detect_func = detect_func
        self.detect_batch_func = detect_batch_func
        self.detect_batch_size = detect_batch_size
        self.detect_smallest = detect_smallest
        self.detect_min_size = detect_min_size
        self.detect_every = detect_every
        self.track_min_confidence = track_min_confidence
        self.track_min_overlap_ratio = track_min_overlap_ratio
        self.track_max_gap = track_max_gap
        self._trackers_backend = trackers
        self.python_lanes = False          # True: the passes of planned shots run in Python too (tests compare the two forms)

    # ---- geometry -------------------------------------------------------------------------------------
    def _match(self, rectangle1, rectangle2):
        overlap = rectangle1.intersect(rectangle2).area()
        if ((overlap < self.track_min_overlap_ratio * rectangle1.area()) or
                (overlap < self.track_min_overlap_ratio * rectangle2.area())):
            overlap = 0.
        return overlap

    def _no_match(self, p1, p2):
        """_match(drectangle(*p1), drectangle(*p2)) == 0 without building the rectangle objects (same float arithmetic:
        intersection corners by max / min, empty rectangles have zero area, the two gating comparisons of tracking.py:129-134)"""
        l1, t1, r1, b1 = float(p1[0]), float(p1[1]), float(p1[2]), float(p1[3])
        l2, t2, r2, b2 = float(p2[0]), float(p2[1]), float(p2[2]), float(p2[3])
        il, it, ir, ib = max(l1, l2), max(t1, t2), min(r1, r2), min(b1, b2)
        overlap = 0.0 if (il > ir or it > ib) else (ir - il) * (ib - it)
        a1 = 0.0 if (l1 > r1 or t1 > b1) else (r1 - l1) * (b1 - t1)
        a2 = 0.0 if (l2 > r2 or t2 > b2) else (r2 - l2) * (b2 - t2)
        ratio = self.track_min_overlap_ratio
        if overlap < ratio * a1 or overlap < ratio * a2:
            return True
        return overlap == 0

    def _associate(self, positions, detections):
        """positions: [(identifier, (l,t,r,b))] in tracker order. Returns {detection index: identifier} (tracking.py:136-182)."""
        if len(positions) < 1 or len(detections) < 1:
            return dict()
        pairs = _lib.associate([p for _, p in positions], detections, self.track_min_overlap_ratio)
        return {d: positions[t][0] for t, d in pairs}

    # ---- one pass over one shot, written as a coroutine that asks for batched tracker work -------------------
    def _lane(self, cache, detections_at, direction, edges, backend=None, plan=None):
        """cache: [(t, frame)] in processing order; detections_at: {t: [box]}; edges: list receiving
        (u, v, confidence) in the order the reference calls add_edge (tracking.py:214-259).
        A coroutine: it yields ('update', handles, frames) / ('start', frames, boxes) and receives the results, so that a
        scheduler can batch the requests of many lanes.  With a speculating backend (HipTrackers.speculate), or a `plan` that
        was computed ahead by another thread, the starts and first updates were already issued in bulk and only trackers that
        survive an association need on-demand updates."""
        if plan is None and backend is not None and hasattr(backend, 'speculate'):
            plan = backend.speculate(cache, detections_at)
        release = backend.release if backend is not None else None
        trackers = {}      # identifier -> backend handle  (dict order == creation order, like the reference's dict)
        position = {}      # identifier -> (l,t,r,b) doubles after the last update
        confidences = {}
        previous = {}
        cached = {}        # identifier -> (psr, box) of a first update computed ahead
        uncommitted = {}   # identifier -> frame of its deferred first update (filters not updated yet)
        deferring = plan is not None and backend is not None and hasattr(backend, 'commit_many')
        new_identifier = 0

        def kill(identifier):
            h = trackers.pop(identifier)
            cached.pop(identifier, None)
            uncommitted.pop(identifier, None)
            if release is not None:
                release(h)
            return h

        for t, frame in cache:
            ids = list(trackers)
            if ids:
                need = [i for i in ids if i not in cached]
                fresh = {}
                late = [i for i in need if i in uncommitted]
                if late:      # survivors of a deferred first update: bring their filters up to date before updating again
                    yield ('commit', [trackers[i] for i in late], [uncommitted.pop(i) for i in late])
                if need:
                    psr, boxes = yield ('update', [trackers[i] for i in need], [frame] * len(need))
                    psr, boxes = _floats(psr), _rows(boxes)          # Python floats, one conversion per batch
                    for k, identifier in enumerate(need):
                        fresh[identifier] = (psr[k], boxes[k])
                for identifier in ids:
                    if deferring and identifier in cached:
                        uncommitted[identifier] = frame
                    conf, pos = cached.pop(identifier) if identifier in cached else fresh[identifier]
                    confidences[identifier] = conf
                    position[identifier] = pos
                    if conf < self.track_min_confidence:
                        h = kill(identifier)
                        if release is None:
                            yield ('release', h)
            detections = detections_at.get(t, [])
            match = self._associate([(i, position[i]) for i in trackers], detections)
            for d, identifier in match.items():
                current = (t, detections[d], DETECTION)
                edges.append((previous[identifier], current, confidences[identifier]))
                h = kill(identifier)
                if release is None:
                    yield ('release', h)
            for identifier in trackers:
                current = (t, position[identifier], direction)
                edges.append((previous[identifier], current, confidences[identifier]))
                previous[identifier] = current
            if detections:
                if plan is not None:
                    handles, ppsr, ppos = plan[t]
                else:
                    handles = yield ('start', [frame] * len(detections), [tuple(float(v) for v in d) for d in detections])
                    ppsr = ppos = None
                if ppsr is not None:
                    ppsr, ppos = _floats(ppsr), _rows(ppos)
                for d, detection in enumerate(detections):
                    trackers[new_identifier] = handles[d]
                    previous[new_identifier] = (t, detection, DETECTION)
                    if ppsr is not None:
                        cached[new_identifier] = (ppsr[d], ppos[d])
                    new_identifier += 1
        for identifier in list(trackers):
            h = kill(identifier)
            if release is None:
                yield ('release', h)

    def _lane_native(self, lane, frames, backend, plan):
        """the same pass run by the library (NativeLane): a coroutine with the protocol of `_lane` -- it only yields when a tracker
        outlives its first update (commit / update requests) -- so LaneScheduler batches the requests of native and Python lanes alike.
        frames: the shot's frames in FORWARD order (the library names frames by their index in the shot)."""
        reply = None
        release = backend.release
        while True:
            req, hs, fs, plan_from = lane.advance(reply)
            reply = None
            for h in lane.take_dead():
                release(h)
            if req == 0:
                return
            if req == 3:
                plan.native_feed(lane, plan_from)
            elif req == 2:
                yield ('commit', hs.tolist(), [frames[i] for i in fs.tolist()])
            else:
                reply = yield ('update', hs.tolist(), [frames[i] for i in fs.tolist()])

    @staticmethod
    def _run_lanes(lanes, backend):
        """Advance all lane coroutines in lock-step; merge their requests into one update batch + one start batch per round."""
        sched = LaneScheduler(backend)
        for g in lanes:
            sched.add(g)
        while len(sched):
            sched.round()

    # ---- merging ----------------------------------------------------------------------------------------
    def _fix(self, track):
        """Merge forward / backward / detection boxes of one timestamp (tracking.py:261-296)."""
        fixed_track = []
        for t, group in itertools.groupby(sorted(track), key=lambda x: x[0]):
            group = list(group)
            if len(group) == 1:
                # the usual case with detection on every frame: one node at this time.  No pair to check, mean of one box = the box
                # (float64 / 1 is exact), int(round()) of its coordinates
                _, pos, status = group[0]
                fixed_track.append((t, (int(round(pos[0])), int(round(pos[1])), int(round(pos[2])), int(round(pos[3]))), status))
                continue
            error = False
            for (_, pos1, _), (_, pos2, _) in itertools.combinations(group, 2):
                if self._no_match(pos1, pos2):
                    error = True
                    break
            status = "+".join(sorted((status for _, _, status in group), key=lambda s: _STATUS_ORDER[s]))
            if error:
                status = "error({0})".format(status)
            # == np.mean(np.vstack(boxes), axis=0): float64 row-by-row sum, then one division -- the same IEEE operations on Python
            # floats; round() of a Python float and of a numpy float64 are both round-half-to-even
            n = len(group)
            pos = []
            for k in range(4):
                s = float(group[0][1][k])
                for g in group[1:]:
                    s = s + float(g[1][k])
                pos.append(int(round(s / n)))
            pos = tuple(pos)
            fixed_track.append((t, pos, status))
        return fixed_track

    def _fill_gaps(self, tracks):
        tracks = sorted(tracks, key=get_min_max_t)
        graph = nx.Graph()
        for i in range(len(tracks)):
            graph.add_node(i)
        for i, j in itertools.combinations(range(len(tracks)), 2):
            ti = tracks[i][-1][0]
            tj = tracks[j][0][0]
            if (tj < ti) or (tj - ti > self.track_max_gap):
                continue
            if not self._no_match(tracks[i][-1][1], tracks[j][0][1]):
                graph.add_edge(i, j)
        merged_tracks = []
        for group in nx.connected_components(graph):
            merged_tracks.append([item for t in sorted(group) for item in tracks[t]])
        return merged_tracks

    def _tracks_from_graph(self, graph):
        timestamps = [t for t in graph if not isinstance(t, tuple)]
        graph.remove_nodes_from(timestamps)
        tracks = nx.connected_components(graph.to_undirected(reciprocal=False, as_view=True))   # same node order, no copy
        tracks = [self._fix(track) for track in tracks]
        tracks = self._fill_gaps(tracks)
        return sorted(tracks, key=get_min_max_t)

    @staticmethod
    def _normalize_track(track, frame_width, frame_height):
        return [(t, (l / frame_width, tp / frame_height, r / frame_width, b / frame_height), status)
                for (t, (l, tp, r, b), status) in track]

    # ---- shots --------------------------------------------------------------------------------------------
    def _detect_shot(self, cache, flags):
        """detections per cached frame (only where flags[i]); batched when a batch function is available"""
        out = [[] for _ in cache]
        idx = [i for i, f in enumerate(flags) if f]
        if self.detect_batch_func is not None:
            bs = max(1, int(self.detect_batch_size))
            for o in range(0, len(idx), bs):
                chunk = idx[o:o + bs]
                res = self.detect_batch_func([cache[i][1] for i in chunk])
                for i, dets in zip(chunk, res):
                    out[i] = [tuple(d) for d in dets]
        else:
            for i in idx:
                out[i] = [tuple(d) for d in self.detect_func(cache[i][1])]
        return out

    def begin_shot(self, cache, flags, dets=None, backend=None, plans=None, det_arrays=None):
        """graph with the detections of one shot + its two lane coroutines (not started); plans = (forward, backward) results of
        HipTrackers.speculate computed ahead (backward: on the reversed cache)"""
        pf, pb = plans if plans is not None else (None, None)
        native = (pf is not None and pb is not None and hasattr(pf, "native_feed") and hasattr(pb, "native_feed") and backend is not None
                  and hasattr(backend, "commit_many") and hasattr(backend, "release") and not self.python_lanes)
        if dets is None and not (native and det_arrays is not None):
            if det_arrays is not None:
                st = np.concatenate([[0], np.cumsum(det_arrays[0])]).tolist()
                bl = np.asarray(det_arrays[1]).astype(np.int64).tolist()
                dets = [[tuple(b) for b in bl[st[i]:st[i + 1]]] for i in range(len(cache))]
            else:
                dets = self._detect_shot(cache, flags)
        if native:
            # both passes' starts and first updates were issued ahead: the passes run in the library (csrc/shotgraph.hip), this thread
            # only serves the trackers that outlive their first update
            counts, boxes = det_arrays if det_arrays is not None else detection_arrays(dets)
            frames = [f for _, f in cache]
            lf = NativeLane(len(cache), counts, boxes, FORWARD, self.track_min_confidence, self.track_min_overlap_ratio)
            lb = NativeLane(len(cache), counts, boxes, BACKWARD, self.track_min_confidence, self.track_min_overlap_ratio)
            return {"times": [t for t, _ in cache], "native": (lf, lb),
                    "lanes": [self._lane_native(lf, frames, backend, pf), self._lane_native(lb, frames, backend, pb)]}
        # the detection graph of the reference (tracking.py:426-429: a node per timestamp, an edge to each of its detections) is kept
        # as what finish_shot needs of it -- its node ORDER -- and built for real only by finish_shot_graph (2000 add_edge calls per
        # 250-frame shot cost 3 ms of the tracking thread, on the critical path of a video's last shot)
        det_at = {}
        for (t, _), d in zip(cache, dets):
            det_at[t] = d
        ef, eb = [], []
        lanes = [self._lane(cache, det_at, FORWARD, ef, backend, pf), self._lane(list(reversed(cache)), det_at, BACKWARD, eb, backend, pb)]
        return {"times": [t for t, _ in cache], "det_at": det_at, "ef": ef, "eb": eb, "lanes": lanes}

    @staticmethod
    def _detection_graph(job):
        g = nx.DiGraph()
        for t in job["times"]:
            g.add_node(t)
            for box in job["det_at"][t]:
                g.add_edge(t, (t, box, DETECTION))
        return g

    @staticmethod
    def _python_view(job):
        """a job whose passes ran in the library, in the shape the Python forms read: the detections per timestamp and the two passes'
        add_edge calls as lists of (u, v, confidence) (tests compare the two forms through it)"""
        if "native" in job and "ef" not in job:
            lf, lb = job["native"]
            times = job["times"]
            starts = np.concatenate([[0], np.cumsum(lf.counts)]).tolist()
            boxes = lf.boxes.astype(np.int64).tolist()          # (detector boxes are integers)
            job["det_at"] = {t: [tuple(b) for b in boxes[starts[i]:starts[i + 1]]] for i, t in enumerate(times)}
            job["ef"], job["eb"] = lf.edges(times), lb.edges(times)
        return job

    def finish_shot_graph(self, job):
        """the reference's own data structure: replay the lanes' graph mutations in its order (forward pass, then backward) into the
        networkx graph and take the tracks from it (tracking.py:359-362)"""
        job = self._python_view(job)
        g = self._detection_graph(job)
        for u, v, conf in job["ef"]:
            g.add_edge(u, v, confidence=conf)
        for u, v, conf in job["eb"]:
            g.add_edge(u, v, confidence=conf)
        return self._tracks_from_graph(g)

    def finish_shot(self, job):
        """Same tracks as finish_shot_graph without building the graph: the tracks are the connected components of the non-timestamp
        nodes, listed in the order of each component's first node in the graph's node order (networkx yields them that way) -- and a
        node enters the graph with the first add_edge that names it: detections frame by frame (begin_shot), then the forward edges,
        then the backward ones.  A union-find over that same node order gives the same components in the same order; what happens to
        them afterwards (_fix, _fill_gaps, the final sort) is shared."""
        if "native" in job:
            lf, lb = job["native"]
            rows, starts = shot_tracks_native(lf, lb, job["times"], self.track_max_gap)
            job["rows"], job["track_start"] = rows, starts
            times = job["times"]
            rl = rows.tolist()
            tracks = []
            for k in range(len(starts) - 1):
                tracks.append([(times[f], (l, tp, r, b), status_of(c)) for f, l, tp, r, b, c in rl[starts[k]:starts[k + 1]]])
            lf.close(); lb.close()
            return tracks
        index = {}
        parent = []
        det_at = job["det_at"]
        for t in job["times"]:                      # node order so far: timestamps and their detections (a box seen twice on a frame is one node)
            for box in det_at[t]:
                n = (t, box, DETECTION)
                if n not in index:
                    index[n] = len(parent)
                    parent.append(len(parent))
        for edges in (job["ef"], job["eb"]):
            for u, v, _ in edges:
                iu = index.get(u)
                if iu is None:
                    iu = index[u] = len(parent)
                    parent.append(iu)
                iv = index.get(v)
                if iv is None:
                    iv = index[v] = len(parent)
                    parent.append(iv)
                while parent[iu] != iu:
                    parent[iu] = parent[parent[iu]]
                    iu = parent[iu]
                while parent[iv] != iv:
                    parent[iv] = parent[parent[iv]]
                    iv = parent[iv]
                if iu != iv:
                    if iu < iv:
                        parent[iv] = iu
                    else:
                        parent[iu] = iv
        comps = {}
        for n, i in index.items():                  # dict order == node order
            while parent[i] != i:
                i = parent[i]
            comps.setdefault(i, []).append(n)
        tracks = [self._fix(track) for track in comps.values()]     # roots are each component's first node: dict order == component order
        tracks = self._fill_gaps(tracks)
        return sorted(tracks, key=get_min_max_t)

    def process_shots(self, shots, backend):
        """shots: list of (cache, flags[, detections]) -- [(t, frame)], [run detection on frame i], optional precomputed
        [[box]] per frame (e.g. from a detector stream running ahead).  Returns one track list per shot.
        All shots' forward and backward passes run in lock-step (they are independent: tracking.py:359-362,410-417)."""
        jobs = [self.begin_shot(s[0], s[1], s[2] if len(s) > 2 else None, backend) for s in shots]
        self._run_lanes([l for j in jobs for l in j["lanes"]], backend)
        return [self.finish_shot(j) for j in jobs]

    @staticmethod
    def _detection_frame(frame, width, height, backend):
        ctx = getattr(backend, "ctx", None)
        if ctx is None:
            raise NotImplementedError("detect_min_size needs frames of the detection size, or a GPU tracker backend that resizes them")
        return ctx.resize(frame, width, height)

    def _backend(self):
        if self._trackers_backend is None:
            from . import runtime
            self._trackers_backend = HipTrackers(runtime.default_context())
        return self._trackers_backend

    def __call__(self, video, segmentation):
        """Yield normalised tracks, shot after shot, in the reference's order (tracking.py:374-434).  With the GPU tracker backend the
        video goes through the streaming engine (engine.py: one reader thread + pinned ingest ring, batched detection, bulk tracker
        work, frames released shot by shot); the tracks are the same, they just exist sooner."""
        every_x_frames = int(self.detect_every * video.frame_rate) if self.detect_every > 0.0 else 1
        if every_x_frames < 1:
            every_x_frames = 1
        width, height = video.size
        ratio = 1.0
        if self.detect_min_size > 0.0:
            ratio = min(1.0, self.detect_smallest / (self.detect_min_size * height))
        old_frame_size = video.frame_size
        frame_width, frame_height = int(width * ratio), int(height * ratio)
        video.frame_size = (frame_width, frame_height)
        backend = self._backend()
        try:
            if isinstance(backend, HipTrackers) and hasattr(backend.ctx, "detect_many"):
                for track in self._call_streaming(video, segmentation, backend, every_x_frames, (width, height), (frame_width, frame_height)):
                    yield track
                return
            segment_generator = get_segment_generator(segmentation)
            segment_generator.send(None)
            cache, flags = [], []
            for i, (t, frame) in enumerate(video):
                # the reference's Video hands out frames already resized to frame_size (video.py:402-403: cv2.resize per frame on the host);
                # a source that yields native frames gets them resized on the device, where they are staged anyway
                if (frame.shape[1], frame.shape[0]) != (frame_width, frame_height):
                    frame = self._detection_frame(frame, frame_width, frame_height, backend)
                segment = segment_generator.send(t)
                if segment:
                    for track in self.process_shots([(cache, flags)], backend)[0]:
                        yield self._normalize_track(track, frame_width, frame_height)
                    cache, flags = [], []
                cache.append((t, frame))
                flags.append(i % every_x_frames == 0)
            for track in self.process_shots([(cache, flags)], backend)[0]:
                yield self._normalize_track(track, frame_width, frame_height)
        finally:
            if self.detect_min_size > 0.0:
                video.frame_size = old_frame_size

    def _call_streaming(self, video, segmentation, backend, every, native_size, detection_size):
        import queue
        import threading
        from . import engine
        ctx = backend.ctx
        out = queue.Queue()
        job = engine.VideoJob(ctx, native_size[0], native_size[1], detection_size[0], detection_size[1], extract=False,
                              on_tracks=lambda tracks: out.put(tracks))
        w, h = native_size
        batch = int(max(8, min(128, 128 * (1920 * 1080) // max(w * h, 1))))
        eng = engine.Engine(ctx, self, detect_batch_size=max(batch, int(self.detect_batch_size)))
        # the engine resizes on the device whatever does not have the detection size yet (checked per shot on the frames themselves:
        # `video.frame_size` was set above, so a reference-style Video already delivers that size; min_size = 0 frames of another size
        # than the announced one are brought to it as the non-streaming path does)
        src = engine.StreamSource(ctx, [(job, video, segmentation, every, detection_size)])
        state = {}

        def work():
            try:
                eng.run(src, backend)
            except BaseException as e:          # noqa: BLE001 -- re-raised in the consumer
                state["error"] = e
            finally:
                out.put(None)

        th = threading.Thread(target=work, name="pvface-tracking")
        th.start()
        try:
            while True:
                tracks = out.get()
                if tracks is None:
                    break
                for track in tracks:
                    yield track
        finally:
            src.close()
            th.join()
        if "error" in state:
            raise state["error"]
