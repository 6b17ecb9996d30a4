"""The streaming engine behind FacePipeline, the CLI verbs and FaceTracking.__call__: shots of one or many videos flow through

    source (decoder thread) -> GPU thread: detect(k), bulk tracker work(k), landmarks + embeddings of finished shots
                            -> caller's thread: association / graph / merging of shot k (tracking.py:184-362), track files, clustering

with memory bounded the way the reference bounds it -- per shot (tracking.py:359-362,410-420: the frame cache is dropped when a shot
has been tracked) -- instead of per video:

  * frames the engine staged itself (numpy frames pushed through the pinned ingest ring, device-side --min-size copies) are released
    as soon as the shot's passes are done and `extract` has walked past them (pvf_frame_release: the buffer is recycled behind an
    event, nothing waits); at most the shots in flight are resident: one being read, one queued, one in detection, one in the state
    machine, one waiting for its faces;
  * the bulk tracker starts of a shot (HipTrackers.speculate_pair) are windowed when the shot holds more detections than
    `speculate_limit`: trackers for the next `speculate_window` detections only, both passes (WindowedPlan).

A "job" is one video: its own track numbering, its own `extract` walk, its own clustering.  Several jobs may share one engine run
(BASELINE.json configs[3]: independent clips farmed to a GPU, no collective): the detector of clip i + 1 runs while the state
machine of clip i is busy on the host.
"""
import os
import queue
import threading
import time as _time

import numpy as np

from . import formats
from .tracking_by_detection import get_segment_generator, HipTrackers


class ListStore(object):
    """frames the caller owns (resident for the whole run): nothing to release"""

    def __init__(self, frames):
        self.frames = frames

    def __getitem__(self, i):
        return self.frames[i]

    def add(self, index, frame, owned):
        pass

    def release_below(self, index):
        pass

    def release_all(self):
        pass


class FrameStore(object):
    """native frames of one video by global frame index; frames the engine staged itself (`owned`) go back to the buffer pool as soon
    as `extract` has passed them"""

    def __init__(self):
        self.frames = {}
        self.owned = set()
        self.low = 0
        self.peak = 0

    def __getitem__(self, i):
        return self.frames[i]

    def add(self, index, frame, owned):
        self.frames[index] = frame
        if owned:
            self.owned.add(index)
        self.peak = max(self.peak, len(self.frames))

    def release_below(self, index):
        for i in range(self.low, index):
            f = self.frames.pop(i, None)
            if f is not None and i in self.owned:
                self.owned.discard(i)
                f.release()
        self.low = max(self.low, index)

    def release_all(self):
        """the job is over: whatever `extract` never reached (the frames behind the last group it hands on) goes back too"""
        for i in sorted(self.owned):
            self.frames[i].release()
        self.owned.clear()
        self.frames.clear()


class ExtractStream(object):
    """`extract` (pyannote-face.py:121-175, 425-466) fed shot by shot while later shots are still being detected.

    The reference reads the finished track file and walks frames and timestamp groups in step (pipeline.faces_per_frame).  Shots
    are disjoint in time and arrive in order, so the same walk can be resumed whenever a shot's tracks exist: groups are
    appended to a queue, the frame pointer only moves while a group is available, and the newest group is held back until a
    later one arrives because the reference's generator never yields the last group of the file.  The faces that become
    available are aligned and embedded immediately.  `frames` is anything indexable by global frame index (a list, a FrameStore);
    `frame_times` may grow while the video streams in."""

    def __init__(self, ctx, frames, frame_times, frame_width, frame_height):
        self.ctx, self.frames, self.times = ctx, frames, frame_times
        self.w, self.h = frame_width, frame_height
        self.tracks, self.rows = [], []
        self.rows_file = []                   # the track file's rows, in file order
        self.file_T, self.file_id = [], []    # ... and their (T, track) columns (they decide the row order of the outputs)
        self.groups, self.gi, self.fi = [], 0, 0
        self.face_boxes, self.face_T, self.face_id = [], [], []
        self.pts, self.emb = [], []
        self.emitted = []     # (frame index, T) of every group handed on, in order

    def _emit(self, available):
        """faces of the groups that may be handed on now: (frames, boxes, frame index below which every face has been handed on)"""
        face_frames, boxes = [], []
        times = self.times
        while self.fi < len(times) and self.gi < available:
            T, g = self.groups[self.gi]
            if T > times[self.fi]:
                self.fi += 1
                continue
            for ident, box in g:
                face_frames.append(self.frames[self.fi]); boxes.append(box)
                self.face_T.append(T); self.face_id.append(ident)
            self.emitted.append((self.fi, T))
            self.gi += 1
            self.fi += 1
        self.face_boxes.extend(boxes)
        return face_frames, boxes, self.fi

    def compute(self, work):
        """GPU part: landmarks + embeddings of one batch of faces returned by prepare(); batches must arrive in order"""
        compute_many(self.ctx, [(self, work)])

    def prepare_rows(self, tracks, shot_times, rows, starts, det_size):
        """prepare() for a shot whose tracks came from the library as arrays (tracking_by_detection.shot_tracks_native: rows = frame,
        l, t, r, b, status code; track k = rows starts[k] .. starts[k + 1]): the same track-file rows and timestamp groups, their numbers
        made by the library in two calls (pvf_round_decimals, pvf_track_rows) instead of 9 Python operations per row"""
        from .tracking_by_detection import status_of
        from . import _lib
        base = len(self.tracks)
        m = len(rows)
        Tq = _lib.round_decimals(np.asarray(shot_times, np.float64)[rows[:, 0]], 3) if m else np.zeros(0)
        q32, pix = _lib.track_rows(rows[:, 1:5], det_size[0], det_size[1], self.w, self.h)
        ids = (base + np.repeat(np.arange(len(starts) - 1), np.diff(starts))).tolist()
        Tl = Tq.tolist()
        boxes = [tuple(b) for b in q32.tolist()]
        status = [status_of(c) for c in rows[:, 5].tolist()]
        self.file_T.extend(Tl)
        self.file_id.extend(ids)
        self.rows_file.extend(zip(Tl, ids, boxes, status))
        self.tracks.extend(tracks)
        order = np.argsort(Tq, kind="stable").tolist()
        pl = [tuple(b) for b in pix.tolist()]
        groups = self.groups
        k = 0
        while k < m:
            T = Tl[order[k]]
            g = []
            while k < m and Tl[order[k]] == T:
                i = order[k]
                g.append((ids[i], pl[i]))
                k += 1
            if groups and groups[-1][0] == T:
                groups[-1][1].extend(g)           # cannot happen for disjoint shots; keeps the grouping rule exact anyway
            else:
                groups.append((T, g))
        return self._emit(len(self.groups) - 1)

    def prepare(self, tracks):
        """host part for the normalised tracks of the next shot (in shot order): the track-file rows, their timestamp groups,
        and the faces that can be extracted now"""
        base = len(self.tracks)
        # the track file's rows: time and box with 3 decimals ('%.3f'), the box then parsed as float32 (pyannote-face.py:125-127).
        # round(float, 3) is the correctly rounded 3-decimal value, i.e. float('%.3f' % v) -- for Python floats only (numpy scalars
        # round differently), hence the float(); the float32 parse is one array cast.
        flat = [round(float(v), 3) for track in tracks for _, box, _ in track for v in box]
        q32 = np.asarray(flat, np.float64).astype(np.float32).astype(np.float64).reshape(-1, 4).tolist() if flat else []
        rows, i = [], 0
        for k, track in enumerate(tracks):
            for t, _, status in track:
                rows.append((round(float(t), 3), base + k, tuple(q32[i]), status))
                i += 1
        self.file_T.extend(r[0] for r in rows)
        self.file_id.extend(r[1] for r in rows)
        self.rows_file.extend(rows)
        rows = sorted(rows, key=lambda r: r[0])
        self.tracks.extend(tracks)
        k, n = 0, len(rows)
        while k < n:
            T = rows[k][0]
            g = []
            while k < n and rows[k][0] == T:
                _, ident, box, _ = rows[k]
                g.append((ident, formats.denormalise(box, self.w, self.h)))
                k += 1
            if self.groups and self.groups[-1][0] == T:
                self.groups[-1][1].extend(g)      # cannot happen for disjoint shots; keeps the grouping rule exact anyway
            else:
                self.groups.append((T, g))
        return self._emit(len(self.groups) - 1)

    def feed(self, tracks):
        self.compute(self.prepare(tracks))

    def plan_finish(self, drop_last=True, reorder=True):
        """Host part of finish() that does not need the embeddings: the last faces to extract, the file order of all faces and the
        sorted track rows.  The pipelined run calls it while the GPU still embeds the last shot's faces."""
        self._final_work = self._emit(len(self.groups) - (1 if drop_last else 0))
        if not drop_last and self.gi < len(self.groups):
            # a shard that is not the end of the video must hand on ALL its groups.  A group is left over when '%.3f' rounded a frame time
            # UP (e.g. 30 fps: t = 0.066667 -> T = 0.067 > t): the reference then serves that group one frame late and carries the lag
            # across the shot boundary, i.e. into the next shard -- a shard cannot reproduce that on its own.  All BASELINE.json
            # configurations run at 25 / 50 fps, whose frame times survive the rounding.
            raise ValueError("frame-range shard ends with %d face group(s) whose rounded time lies behind the shard's last frame; cut the "
                             "video at shots whose frame times survive 3-decimal rounding (25 / 50 fps do) or run it unsharded"
                             % (len(self.groups) - self.gi))
        self._perm = None
        if reorder and len(self.face_T):
            perm = formats.file_order(self.face_T, self.face_id, self.file_T, self.file_id)
            self._perm = perm
            self.face_T = [self.face_T[i] for i in perm]
            self.face_id = [self.face_id[i] for i in perm]
            self.face_boxes = [self.face_boxes[i] for i in perm]
        rows_file = self.rows_file
        self.rows = [rows_file[i] for i in formats.pandas_sort_order(self.file_T).tolist()]
        self._early = None
        if self._perm is not None:
            # the rows computed so far move to their places in file order now, beside the GPU's last batch: finish() then only places
            # that batch (a quarter of the 8.4 MB of a configs[1] step; the whole move used to sit at the step's very end, GPU idle).
            # The GPU thread may be appending: `emb` is appended after `pts`, so its length counts the complete batches.
            k0 = len(self.emb)
            perm = np.asarray(self._perm, np.int64)
            n = len(perm)
            m0 = sum(len(e) for e in self.emb[:k0])
            fp, fe = np.empty((n, 68, 2), np.int32), np.empty((n, 128), np.float32)
            early = perm < m0
            if m0:
                src = perm[early]
                fp[early] = np.concatenate(self.pts[:k0])[src]
                fe[early] = np.concatenate(self.emb[:k0])[src]
            self._early = (k0, m0, perm, early, fp, fe)
        self._planned = True

    def finish(self, drop_last=True, reorder=True, computed=False):
        """reorder: put the faces of one timestamp into the order the reference's `extract` writes them (formats.file_order).
        A shard of a longer video leaves that to the step that sees the whole track table (dist.gather_rows).
        computed: the engine's GPU thread has already run the final batch (plan_finish()'s work)."""
        if not getattr(self, "_planned", False):
            self.plan_finish(drop_last, reorder)
        if not computed:
            self.compute(self._final_work)
        if getattr(self, "_early", None) is not None:
            k0, m0, perm, early, fp, fe = self._early
            # every row of fp / fe must have a source: the rows placed early plus the batches appended since (a final batch that was
            # skipped or failed would leave np.empty rows behind -- ADVICE r5)
            if sum(len(e) for e in self.emb) != len(perm) or len(self.pts) != len(self.emb):
                raise RuntimeError("extraction store: %d descriptor rows computed for %d faces planned (the final batch did not run?)"
                                   % (sum(len(e) for e in self.emb), len(perm)))
            if len(self.pts) > k0:
                late = ~early
                src = perm[late] - m0
                fp[late] = (np.concatenate(self.pts[k0:]) if len(self.pts) > k0 + 1 else self.pts[k0])[src]
                fe[late] = (np.concatenate(self.emb[k0:]) if len(self.emb) > k0 + 1 else self.emb[k0])[src]
            return fp, fe
        pts = np.concatenate(self.pts) if self.pts else np.zeros((0, 68, 2), np.int32)
        emb = np.concatenate(self.emb) if self.emb else np.zeros((0, 128), np.float32)
        if self._perm is not None:
            pts, emb = pts[self._perm], emb[self._perm]
        return pts, emb


EXTRACT_CALL_MAX = 4096          # faces per landmark / embedding call (the network's largest forward)


def compute_many(ctx, items):
    """landmarks + embeddings of several batches -- [(ExtractStream, work)], possibly of different videos -- in ONE library call (a call
    costs 2-4 ms of idle GPU around its kernels whatever its size, and the deep layers of the network fill the chip only from a
    few thousand faces on); every stream receives its own rows, in order"""
    frames, boxes, cuts = [], [], []
    for ex, work in items:
        if work is not None and work[1]:
            frames.extend(work[0]); boxes.extend(work[1])
            cuts.append((ex, len(work[1])))
    if not boxes:
        return
    if hasattr(ctx, "landmarks_embed"):
        pts, emb = ctx.landmarks_embed(frames, boxes)      # one library call: no interpreter between the two stages
    else:
        pts = ctx.landmarks(frames, boxes)
        emb = ctx.embed(frames, pts)
    if len(cuts) == 1:
        cuts[0][0].pts.append(pts); cuts[0][0].emb.append(emb)
        return
    a = 0
    for ex, m in cuts:
        ex.pts.append(pts[a:a + m]); ex.emb.append(emb[a:a + m])
        a += m


def detection_arrays_of(n_frames, raw):
    """(counts int32 [n_frames], boxes float64 [sum, 4]) from the arrays of Context.detect_many(arrays=True): what the library's passes take"""
    counts = np.zeros(n_frames, np.int32)
    if raw is None:
        return counts, np.zeros((0, 4), np.float64)
    out, cnt, idx = raw
    counts[idx] = cnt
    return counts, out[np.arange(out.shape[1])[None, :] < np.asarray(cnt)[:, None]].astype(np.float64)


def detections_as_lists(n_frames, raw):
    """[[(l, t, r, b) Python ints]] per frame from the arrays of Context.detect_many(arrays=True): raw = (boxes, counts, frame indices)"""
    dets = [[] for _ in range(n_frames)]
    if raw is not None:
        out, cnt, idx = raw
        rows, cnt = out.tolist(), cnt.tolist()
        for j, i in enumerate(idx):
            dets[i] = [tuple(b) for b in rows[j][:cnt[j]]]
    return dets


class ShotInput(object):
    """one shot of one job on its way through the engine"""
    __slots__ = ("job", "base", "cache", "flags", "natives", "owned", "resize")

    def __init__(self, job, base, cache, flags, natives=None, owned=False, resize=None):
        self.job, self.base, self.cache, self.flags = job, base, cache, flags
        self.natives = natives          # native frames when the cache holds (or will hold) down-scaled detection frames
        self.owned = owned              # the native frames were staged by the engine: released when extract has passed them
        self.resize = resize            # (width, height) of the detection frames still to be made on the device (--min-size)


def release_shot_frames(si):
    """error / shutdown paths: a shot that will not be processed gives back the frames the engine staged for it (frames the caller
    owns are left alone; a frame released twice is reported by the library and ignored here)"""
    if not isinstance(si, ShotInput):
        return
    frames = []
    if si.natives is not None:
        frames.extend(f for _, f in si.cache)                 # down-scaled copies the engine made
        if si.owned:
            frames.extend(si.natives)
    elif si.owned:
        frames.extend(f for _, f in si.cache)
    else:
        frames.extend(f for _, f in si.cache if getattr(f, "transient", False))
    for f in frames:
        try:
            f.release()
        except Exception:       # noqa: BLE001 -- already released (the store got there first)
            pass


class JobEnd(object):
    """marks the end of a job's shots in the source"""
    __slots__ = ("job",)

    def __init__(self, job):
        self.job = job


class VideoJob(object):
    """one video going through the engine"""

    def __init__(self, ctx, width, height, det_width=None, det_height=None, frames=None, times=None, extract=True, last_shard=True,
                 reorder=True, on_tracks=None, key=None):
        self.ctx, self.key = ctx, key
        self.w, self.h = int(width), int(height)
        self.tw, self.th = int(det_width or width), int(det_height or height)
        self.store = ListStore(frames) if frames is not None else FrameStore()
        self.times = times if times is not None else []
        self.streaming = times is None          # frame times (and frames) arrive with the shots
        self.ex = ExtractStream(ctx, self.store, self.times, self.w, self.h) if extract else None
        self.tracks = []                        # normalised tracks in yield order (the ExtractStream keeps its own list)
        self.last_shard, self.reorder, self.on_tracks = last_shard, reorder, on_tracks
        self.final_computed = threading.Event()
        self.shot_ranges = []
        self.t_tracked = None                   # when the job's last shot had its tracks
        self.result = None

    def accept(self, si):
        """host thread, when a shot's detections arrive: the job learns the shot's frames and times (streaming sources)"""
        n = len(si.cache)
        self.shot_ranges.append((si.base, si.base + n))
        if self.streaming:
            natives = si.natives if si.natives is not None else [f for _, f in si.cache]
            for j, f in enumerate(natives):
                self.store.add(si.base + j, f, si.owned or getattr(f, "transient", False))
            self.times.extend(t for t, _ in si.cache)

    def shot_tracked(self, si, tracks, normalize, native=None):
        """host thread: the shot's tracks exist.  Returns the extraction work of the faces that may be computed now (or None).
        native: (rows, track_start) when the tracks came from the library as arrays (the same tracks; their numbers are then made there)"""
        if native is not None:
            rows, starts = native
            # == normalize(): box / detection size in float64 (one array division instead of four per row)
            nb = (rows[:, 1:5].astype(np.float64) / np.array([self.tw, self.th, self.tw, self.th], np.float64)).tolist()
            norm, i = [], 0
            for tr in tracks:
                norm.append([(t, tuple(nb[i + j]), st) for j, (t, _, st) in enumerate(tr)])
                i += len(tr)
        else:
            norm = [normalize(tr, self.tw, self.th) for tr in tracks]
        if self.on_tracks is not None:
            self.on_tracks(norm)
        self.t_tracked = _time.perf_counter()
        if self.ex is None:
            self.tracks.extend(norm)
            if self.streaming:
                self.store.release_below(si.base + len(si.cache))
            return None
        if native is not None:
            return self.ex.prepare_rows(norm, [t for t, _ in si.cache], native[0], native[1], (self.tw, self.th))
        return self.ex.prepare(norm)


class _NoLock(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class FairLock(object):
    """first come, first served (threading.Lock hands itself to whoever runs next: a thread that releases it and asks again at once keeps
    it, and the GPU thread's back-to-back library calls would starve the tracking thread's on-demand calls for whole shots)"""

    def __init__(self):
        self._c = threading.Condition(threading.Lock())
        self._next = self._serving = 0
        self._gone = set()          # tickets whose holder gave up while waiting (an exception, e.g. KeyboardInterrupt): skipped when served

    def acquire(self):
        with self._c:
            me = self._next
            self._next += 1
            try:
                while self._serving != me:
                    self._c.wait()
            except BaseException:
                # the ticket must not block everybody behind it: served already -> pass it on, otherwise mark it to be skipped
                if self._serving == me:
                    self._advance()
                else:
                    self._gone.add(me)
                raise

    def _advance(self):
        self._serving += 1
        while self._serving in self._gone:
            self._gone.discard(self._serving)
            self._serving += 1
        self._c.notify_all()

    def release(self):
        with self._c:
            self._advance()

    def waiting(self):
        return self._next - self._serving - 1

    def __enter__(self):
        self.acquire()
        return self

    def __exit__(self, *exc):
        self.release()


class _InterpreterTuning(object):
    """Process-wide interpreter settings a pipelined run wants, reference-counted over the engines that are running (two engines at
    once used to restore each other's values in the wrong order and could leave the 0.1 ms switch interval behind for good):
      * no cyclic garbage collection -- a full collection in the middle of a shot stalls both threads for tens of milliseconds;
      * a 0.1 ms thread switch interval -- the GPU thread re-takes the interpreter lock after every library call, and with the default
        5 ms each of those hand-overs can stall the GPU queue for milliseconds while the caller's thread runs the tracking state machine.
    The first engine in sets them, the last one out restores what it found.  They are in effect in the caller's code too while a
    generator-backed run (TrackingByDetection.__call__) is suspended between two yields; PVF_NO_INTERPRETER_TUNING=1 leaves both alone."""

    def __init__(self):
        self._mu = threading.Lock()
        self._users = 0
        self._saved = None

    def __enter__(self):
        if os.environ.get("PVF_NO_INTERPRETER_TUNING") == "1":
            return self
        import gc
        import sys
        with self._mu:
            if self._users == 0:
                self._saved = (gc.isenabled(), sys.getswitchinterval())
                gc.disable()
                sys.setswitchinterval(1e-4)
            self._users += 1
        return self

    def __exit__(self, *exc):
        if os.environ.get("PVF_NO_INTERPRETER_TUNING") == "1":
            return False
        import gc
        import sys
        with self._mu:
            self._users -= 1
            if self._users == 0 and self._saved is not None:
                was_enabled, interval = self._saved
                self._saved = None
                sys.setswitchinterval(interval)
                if was_enabled:
                    gc.enable()
        return False


_interpreter_tuning = _InterpreterTuning()


class WindowedPlan(object):
    """plan[t] of one pass over a shot (what HipTrackers.speculate_pair returns for the whole shot at once), computed for a window
    of detections at a time when the lane gets there: the trackers that exist at any moment are those of the windows the two lanes
    are in, not those of the whole shot.  Frames, counts and boxes are in the pass's processing order."""

    def __init__(self, backend, frame_handles, times, counts, boxes, window):
        self.backend, self.fh, self.counts, self.boxes, self.window = backend, frame_handles, counts, boxes, max(int(window), 1)
        self.index = {t: i for i, t in enumerate(times)}
        self.times = times
        self.starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.hi = 0
        self.cur = {}
        self.windows = 0

    def __getitem__(self, t):
        i = self.index[t]
        while i >= self.hi:
            self._advance()
        return self.cur.pop(t)

    def _window(self):
        """the next window's frames [lo, hi) and its trackers' arrays (None: no detection in it)"""
        lo, hi, n = self.hi, self.hi, len(self.counts)
        total = 0
        while hi < n and (hi == lo or total + self.counts[hi] <= self.window):
            total += self.counts[hi]
            hi += 1
        self.hi = hi
        self.windows += 1
        if total == 0:
            return lo, hi, None
        k0, k1 = int(self.starts[lo]), int(self.starts[hi])
        owner = np.repeat(np.arange(lo, hi), self.counts[lo:hi])
        return lo, hi, self.backend.speculate_window(self.fh, owner, self.boxes[k0:k1], n)

    def native_feed(self, lane, p):
        """a pass run by the library (tracking_by_detection.NativeLane) asks for the plan from processing frame p on: the next window"""
        assert p >= self.hi, (p, self.hi)
        n = len(self.counts)
        while self.hi <= p:                                  # (windows of frames without a detection are never asked for: passed over here)
            lo, hi, arrays = self._window()
            has = np.arange(lo, hi) != n - 1
            if arrays is None:
                lane.feed(lo, has, np.zeros(0, np.uint64), np.zeros(0), np.zeros((0, 4)))
            else:
                lane.feed(lo, has, arrays[0], arrays[1], arrays[2])

    def _advance(self):
        lo, hi, arrays = self._window()
        n = len(self.counts)
        if arrays is None:
            return
        k0 = int(self.starts[lo])
        hs, psr, pos = arrays
        hl = hs.tolist()
        for i in range(lo, hi):
            m = int(self.counts[i])
            if m:
                a = int(self.starts[i]) - k0
                last = (i == n - 1)
                self.cur[self.times[i]] = (hl[a:a + m], None if last else psr[a:a + m], None if last else pos[a:a + m])


class _LaneBackend(object):
    """what the lanes of the tracking thread see of the tracker context while the GPU thread owns it: on-demand updates
    take the context lock, killed trackers are only queued (the GPU thread destroys them between its batches)"""

    def __init__(self, backend, lock, dead, note=None):
        self.backend, self.lock, self.dead = backend, lock, dead
        self.note = note or (lambda *ev: None)

    def update_many(self, handles, frames):
        with self.lock:
            return self.backend.update_many(handles, frames)

    def commit_many(self, handles, frames):
        with self.lock:
            self.backend.commit_many(handles, frames)

    def start_many(self, frames, boxes):
        with self.lock:
            return self.backend.start_many(frames, boxes)

    def speculate_window(self, fh, owner, boxes, n_frames):
        self.note("window wanted", len(owner))
        with self.lock:
            self.note("window begin")
            self.release_dead()          # the killed trackers of the previous windows make room first
            try:
                return self.backend.speculate_window(fh, owner, boxes, n_frames)
            finally:
                self.note("window done")

    def release_dead(self):
        batch = []
        while self.dead:
            batch.append(self.dead.pop())
        if batch:
            self.backend.release_many(batch)

    def release(self, handle):
        self.dead.append(handle)


class Engine(object):
    """One pass of a source of ShotInput / JobEnd items through the GPU thread and the caller's thread (see the module text)."""

    def __init__(self, ctx, tracking, detect_batch_size=8, overlap=True, speculate_limit=8192, speculate_window=4096, group=1, extract_min=0):
        self.ctx, self.tracking = ctx, tracking
        # shots whose passes run in lock-step as ONE set of lanes (tracking.py:359-362: shots are independent).  With detection on every
        # frame a shot's trackers live for one frame and the bulk calls cover them; with `--every` they live for many frames, updated
        # frame after frame in small batches -- a latency-bound chain per shot -- and several shots side by side share each of those calls.
        self.group = max(1, int(group))
        self.detect_batch_size = detect_batch_size
        self.overlap = overlap
        self.speculate_limit, self.speculate_window = int(speculate_limit), int(speculate_window)
        # faces that must be waiting before the GPU thread runs an extraction call while it still has shots to detect (0: run what
        # is there).  A run over many short videos sets it: their per-shot and end-of-video batches are small, and a call's fixed
        # cost is paid per call (compute_many)
        self.extract_min = int(extract_min)
        self.stats = {}

    # ---- the GPU-side work of one shot ---------------------------------------------------------------------------------------
    def _detect(self, si, lock=None):
        """detections of a shot's flagged frames.  lock: the engine's context lock (pipelined run).  A shot of four or more detector
        batches (4K frames: 32 per batch) is detected batch by batch with the lock given up in between, so that the tracking thread's
        bulk tracker calls for the PREVIOUS shot are served while this one is still being detected (a whole-shot call holds the context
        for a quarter of a second there); shorter shots go through in one call, whose batches hide each other's host work."""
        ctx = self.ctx
        if lock is None:
            lock = _NoLock()
        with lock:
            if si.resize is not None:
                # decided from the frames themselves: a source that honours `frame_size` (the reference's Video resizes on the host,
                # video.py:402-403) already delivers detection-size frames and gets no second copy; native frames are resized here
                tw, th = si.resize
                if any((int(f.shape[1]), int(f.shape[0])) != (tw, th) for _, f in si.cache):
                    # built aside and put in place only when every copy exists: should a resize fail half way, the copies made so far go
                    # back here and `natives` / `cache` still say what they said before (release_shot_frames reads them) -- ADVICE r5
                    resized = []
                    try:
                        for t, f in si.cache:
                            resized.append((t, ctx.resize(f, tw, th)))
                    except BaseException:
                        for _, f in resized:
                            try:
                                f.release()
                            except Exception:       # noqa: BLE001 -- the first error is the one to report
                                pass
                        raise
                    si.natives, si.cache = [f for _, f in si.cache], resized
                si.resize = None
        cache, flags = si.cache, si.flags
        idx = [i for i, f in enumerate(flags) if f]
        self.stats["frames_detected"] = self.stats.get("frames_detected", 0) + len(idx)      # frames that go through the detector kernels
        counts = np.zeros(len(cache), np.int64)
        boxes = np.zeros((0, 4), np.float64)
        raw = None
        if idx:
            batch = max(1, int(self.detect_batch_size))
            chunk = batch if len(idx) >= 4 * batch else len(idx)
            outs, cnts = [], []
            for o in range(0, len(idx), chunk):
                with lock:
                    out, _, cnt = ctx.detect_many([cache[i][1] for i in idx[o:o + chunk]], batch, 1, arrays=True)
                outs.append(out); cnts.append(cnt)
            if len(outs) == 1:
                out, cnt = outs[0], cnts[0]
            else:
                m = max(o.shape[1] for o in outs)
                out = np.concatenate([np.pad(o, ((0, 0), (0, m - o.shape[1]), (0, 0))) for o in outs])
                cnt = np.concatenate(cnts)
            # the boxes go back to the GPU (tracker starts) as an array; the tracking thread turns them into the Python
            # tuples its state machine works on while the GPU is busy with those starts
            counts[idx] = cnt
            boxes = out[np.arange(out.shape[1])[None, :] < cnt[:, None]].astype(np.float64)
            raw = (out, cnt, idx)
        return raw, counts, boxes

    def _speculate(self, si, backend, lane_backend, raw, counts, boxes):
        cache = si.cache
        n = int(counts.sum())
        if hasattr(backend, "speculate_pair"):
            if n <= self.speculate_limit or not hasattr(backend, "speculate_window"):
                return backend.speculate_pair(cache, None, counts=counts, boxes=boxes)
            # a shot with more detections than the tracker budget: both passes ask for their trackers window by window
            fh = self.ctx.frame_handles([f for _, f in cache])
            times = [t for t, _ in cache]
            cnt = np.asarray(counts, np.int64)
            starts = np.concatenate([[0], np.cumsum(cnt)])
            rev = np.concatenate([boxes[starts[i]:starts[i + 1]] for i in range(len(cnt) - 1, -1, -1)]).reshape(-1, 4) if n else boxes
            self.stats["windowed_shots"] = self.stats.get("windowed_shots", 0) + 1
            return (WindowedPlan(lane_backend, fh, times, cnt, boxes, self.speculate_window),
                    WindowedPlan(lane_backend, fh[::-1].copy(), times[::-1], cnt[::-1].copy(), rev, self.speculate_window))
        det_at = {t: d for (t, _), d in zip(cache, detections_as_lists(len(cache), raw))}
        return backend.speculate(cache, det_at), backend.speculate(list(reversed(cache)), det_at)

    # ---- sequential form (no GPU-feeding thread): every stage in the caller's thread, shot after shot --------------------------
    def _run_sequential(self, source, backend):
        jobs = []
        for item in source:
            if isinstance(item, JobEnd):
                job = item.job
                if job.ex is not None:
                    job.ex.plan_finish(drop_last=job.last_shard, reorder=job.reorder)
                    job.ex.compute(job.ex._final_work)
                job.store.release_all()
                job.final_computed.set()
                jobs.append(job)
                continue
            si = item
            raw, counts, boxes = self._detect(si)
            si.job.accept(si)
            plans = self._speculate(si, backend, backend, raw, counts, boxes)
            jb = self.tracking.begin_shot(si.cache, si.flags, None, backend, plans, det_arrays=detection_arrays_of(len(si.cache), raw))
            self.tracking._run_lanes(jb["lanes"], backend)
            tracks = self.tracking.finish_shot(jb)
            self._release_detection_frames(si)
            work = si.job.shot_tracked(si, tracks, self.tracking._normalize_track, (jb["rows"], jb["track_start"]) if "rows" in jb else None)
            if work is not None:
                si.job.ex.compute(work)
                si.job.store.release_below(work[2])
        return jobs

    @staticmethod
    def _release_detection_frames(si):
        if si.natives is not None:              # the cache holds down-scaled copies made by the engine: the passes are done with them
            for _, f in si.cache:
                f.release()

    # ---- pipelined form ----------------------------------------------------------------------------------------------------------
    def run(self, source, backend=None, n_shots=None, on_job_final=None):
        """source: iterable of ShotInput / JobEnd (a job's JobEnd after its last shot).  n_shots: number of shots when known in advance
        (list inputs): enables the end-of-run ordering that keeps the GPU busy during the last shot's host phase.
        on_job_final(job): called in the caller's thread once a job's last faces have been computed, while later jobs are still running.
        Returns the jobs in the order they ended."""
        backend = backend if backend is not None else HipTrackers(self.ctx)
        if not self.overlap:
            jobs = self._run_sequential(source, backend)
            if on_job_final is not None:
                for job in jobs:
                    on_job_final(job)
            return jobs
        with _interpreter_tuning:
            return self._run_pipelined(source, backend, n_shots, on_job_final)

    def _run_pipelined(self, source, backend, n, on_job_final):
        """GPU thread: detect(k), speculate(k), extract(k-1) ...; this thread: lanes + merging of shot k as soon as its detections
        and bulk tracker results exist.  ctypes releases the GIL inside every library call."""
        ready, done = queue.Queue(), queue.Queue()
        lock = FairLock()
        dead = []
        trace = [(_time.perf_counter(), "begin")] if os.environ.get("PVF_TRACE") else None

        def note(*ev):
            if trace is not None:
                trace.append((_time.perf_counter(),) + ev)

        lane_backend = _LaneBackend(backend, lock, dead, note)

        pending = []                                # messages of the tracking thread whose faces have not been computed yet, in order

        def faces_of(msg):
            kind, job, work = msg
            if kind == "final":
                work = job.ex._final_work if job.ex is not None else None
            return work

        def receive(msg, counters):
            pending.append(msg)
            if msg[0] == "work":
                counters["received"] += 1
                ahead.release()                      # the shot's tracks exist: the detector may take another one on

        def run_pending(counters, force):
            """the faces that are waiting go through the landmark / embedding kernels, <= 4096 per library call (the network's largest
            forward), whichever shots and videos they belong to (compute_many); unless `force`, only once extract_min faces wait"""
            while pending:
                if not force and sum(len(w[1]) for w in map(faces_of, pending) if w is not None) < self.extract_min:
                    return
                take, faces = [], 0
                while pending:
                    w = faces_of(pending[0])
                    m = len(w[1]) if w is not None else 0
                    if take and faces + m > EXTRACT_CALL_MAX:
                        break
                    take.append(pending.pop(0))
                    faces += m
                note("extract begin", counters["extracted"])
                if faces:
                    # one library call per <= 4096 faces, the context given up in between (a crowded shot has 10 000: the tracking
                    # thread's bulk tracker calls for the next shot are served between the pieces instead of after 80 ms)
                    items = [(msg[1].ex, faces_of(msg)) for msg in take if msg[1].ex is not None and faces_of(msg) is not None]
                    piece, m = [], 0
                    for ex, w in items:
                        a, n_w = 0, len(w[1])
                        while a < n_w:
                            b = min(n_w, a + EXTRACT_CALL_MAX - m)
                            piece.append((ex, (w[0][a:b], w[1][a:b])))
                            m += b - a
                            a = b
                            if m >= EXTRACT_CALL_MAX:
                                with lock:
                                    compute_many(self.ctx, piece)
                                piece, m = [], 0
                    if piece:
                        with lock:
                            compute_many(self.ctx, piece)
                for kind, job, work in take:
                    if kind == "work":
                        if work is not None:
                            job.store.release_below(work[2])
                        counters["extracted"] += 1
                    else:                           # "final": the job's last faces (plan_finish has run on the host)
                        job.store.release_all()
                        counters["finals"] += 1
                        job.final_computed.set()
                        ready.put(("final done", job))
                note("extracted", counters["extracted"] - 1)

        # ---- two GPU-feeding threads (round 4), one per stream of the context:
        #   detector thread   detect(k + 1): pyramids, FHOG, scoring, NMS on the detector's stream and lock -- needs nothing but frames
        #   tracker thread    speculate(k): the bulk tracker starts + first updates; landmarks + descriptors of finished shots; on the
        #                     context's main stream, whose entry points it shares (FairLock) with the on-demand tracker calls of the
        #                     tracking thread.
        # The detector of shot k + 1 (VALU- / MFMA-bound, fills the chip) runs BESIDE the latency-bound tracker and extraction work of
        # shot k instead of after it (reference tracking.py:199-259 vs :426: independent across shots).
        # shots that are detected (or being detected) and whose tracks do not exist yet -- their frames are resident: one in the detector,
        # one in the bulk tracker work, one in the state machine (groups of shots: two groups)
        limit = max(3, 2 * self.group)
        ahead = threading.Semaphore(limit)
        stop = threading.Event()
        waiting = []                                 # the tracker thread's detections / job ends not yet handled (the error path empties it)

        def detector_thread():
            k = 0
            item = None                     # the shot in this thread's hands: handed on with ("det", ...) or given back on the way out
            try:
                for item in source:
                    if stop.is_set():
                        release_shot_frames(item)
                        return
                    if isinstance(item, JobEnd):
                        done.put(("jobend", item.job))
                        item = None
                        continue
                    while not ahead.acquire(timeout=0.05):
                        if stop.is_set():
                            release_shot_frames(item)
                            return
                    note("detect begin", k)
                    raw, counts, boxes = self._detect(item, None)
                    note("detected", k)
                    done.put(("det", item, raw, counts, boxes))
                    item = None
                    k += 1
                done.put(("eof",))
            except BaseException as e:      # noqa: BLE001 -- handed on to the caller's thread through the tracker thread
                # the shot that was being detected (with the device-resized copies _detect made for it) goes back to the pool here:
                # nobody else has seen it, and cyclic GC is off during a run
                try:
                    release_shot_frames(item)
                except Exception:           # noqa: BLE001 -- the first error is the one to report
                    pass
                done.put(("error", e))

        def gpu_thread():
            counters = {"received": 0, "extracted": 0, "finals": 0}
            shots = ends = 0
            eager = self.extract_min <= 0
            group = []
            eof = False

            def flush():
                if group:
                    ready.put(("shots", list(group)))
                    del group[:]

            def speculate(msg):
                nonlocal shots
                _, si, raw, counts, boxes = msg
                k = shots
                if group and group[-1][0].job is not si.job:
                    flush()
                note("speculate begin", k)
                with lock:
                    lane_backend.release_dead()
                    plans = self._speculate(si, backend, lane_backend, raw, counts, boxes)
                note("speculated", k)
                shots += 1
                group.append((si, raw, plans))
                if len(group) >= self.group or (n is not None and k == n - 1):
                    flush()

            def faces_waiting():
                return sum(len(w[1]) for w in map(faces_of, pending) if w is not None)

            try:
                while True:
                    # what this thread could do now.  Bulk tracker work first (the tracking thread waits for it), but never more than
                    # `limit` shots ahead of that thread (a slow state machine -- a crowded shot -- must not let detected shots pile up).
                    can_spec = bool(waiting) and (waiting[0][0] == "jobend" or shots - counters["received"] < limit)
                    last_call = eof and not waiting                  # nothing will be detected any more: whatever waits goes, whatever its size
                    # Towards the end of a run of known length the faces of the last finished shots are held back until the LAST shot's
                    # bulk tracker work is queued: its state machine (17-21 ms on the host, plus its on-demand tracker calls) then runs
                    # beside ~25 ms of landmark / embedding kernels instead of leaving the GPU idle at the very end.
                    hold = n is not None and shots < n and counters["extracted"] >= n - 3 and not last_call
                    can_extract = bool(pending) and not hold and (last_call or eager or faces_waiting() >= self.extract_min)
                    if last_call:
                        flush()
                        if not pending and counters["extracted"] >= shots and counters["finals"] >= ends:
                            break
                    try:
                        msg = done.get_nowait() if (can_spec or can_extract) else done.get()
                    except queue.Empty:
                        msg = False
                    if msg is None:
                        return
                    if msg is not False:
                        kind = msg[0]
                        if kind == "error":
                            raise msg[1]
                        if kind in ("det", "jobend"):
                            waiting.append(msg)                      # (a job's end keeps its place behind the job's last shot)
                        elif kind == "eof":
                            eof = True
                        else:
                            receive(msg, counters)
                        continue                                     # look again: the state has changed
                    if can_spec:
                        w = waiting.pop(0)
                        if w[0] == "jobend":
                            flush()
                            ends += 1
                            ready.put(("end", w[1]))
                        else:
                            speculate(w)
                        continue
                    if can_extract:
                        run_pending(counters, True)                  # (the decision was taken above)
                flush()
                ready.put(("stop",))
                with lock:
                    lane_backend.release_dead()
                ready.put(("idle",))
            except BaseException as e:      # noqa: BLE001 -- handed to the caller's thread
                stop.set()
                ready.put(e)

        det_th = threading.Thread(target=detector_thread, name="pvface-detector")
        det_th.start()
        th = threading.Thread(target=gpu_thread, name="pvface-gpu")
        th.start()
        finished, ok = [], False
        seen_jobs = []
        k = 0
        try:
            while True:
                item = ready.get()
                if isinstance(item, BaseException):
                    raise item
                kind = item[0]
                if kind == "idle":
                    break
                if kind == "stop":
                    continue
                if kind == "end":
                    job = item[1]
                    if job.ex is not None:
                        job.ex.plan_finish(drop_last=job.last_shard, reorder=job.reorder)     # runs while the GPU embeds the last faces
                    note("planned")
                    done.put(("final", job, None))
                    continue
                if kind == "final done":
                    job = item[1]
                    finished.append(job)
                    if on_job_final is not None:
                        on_job_final(job)
                    continue
                members = item[1]
                note("host begin", k)
                jbs = []
                for si, raw, plans in members:
                    if si.job not in seen_jobs:
                        seen_jobs.append(si.job)
                    si.job.accept(si)
                    jbs.append(self.tracking.begin_shot(si.cache, si.flags, None, lane_backend, plans, det_arrays=detection_arrays_of(len(si.cache), raw)))
                self.tracking._run_lanes([lane for jb in jbs for lane in jb["lanes"]], lane_backend)
                note("lanes done", k)
                for (si, _, _), jb in zip(members, jbs):
                    tracks = self.tracking.finish_shot(jb)
                    note("tracked", k)
                    self._release_detection_frames(si)
                    native = (jb["rows"], jb["track_start"]) if "rows" in jb else None
                    done.put(("work", si.job, si.job.shot_tracked(si, tracks, self.tracking._normalize_track, native)))
                    note("prepared", k)
                    k += 1
            ok = True
        finally:
            if not ok:
                stop.set()
                done.put(None)
            th.join()
            det_th.join()
            if not ok:
                # an error on either side: the frames the engine staged go back to the pool now, not whenever the garbage collector (which
                # a run switches off) finds their handles: shots still queued for this thread, then everything the jobs' stores hold
                while True:
                    try:
                        item = ready.get_nowait()
                    except queue.Empty:
                        break
                    if isinstance(item, tuple) and item and item[0] == "shots":
                        for si, _, _ in item[1]:
                            release_shot_frames(si)
                while True:                          # detected shots nobody took on (both feeding threads have ended)
                    try:
                        waiting.append(done.get_nowait())
                    except queue.Empty:
                        break
                for msg in waiting:
                    if isinstance(msg, tuple) and msg and msg[0] == "det":
                        release_shot_frames(msg[1])
                del waiting[:]
                for job in seen_jobs:
                    try:
                        job.store.release_all()
                    except Exception:       # noqa: BLE001 -- the original error is the one to report
                        pass
                try:
                    self.ctx.pool_trim(0)
                except Exception:           # noqa: BLE001 -- (scripted test contexts have no pool)
                    pass
        while not ready.empty():
            item = ready.get()
            if isinstance(item, BaseException):
                raise item
        if trace is not None:
            import json
            note("finish")
            with open(os.environ["PVF_TRACE"], "w") as f:
                json.dump(trace, f)
        return finished


# ---- sources ----------------------------------------------------------------------------------------------------------------------
def split_into_shots(times, shots):
    """frame index ranges per shot, using the reference's flush rule: a frame at t >= segment.end opens the next shot
    (tracking.py:44-58,406-417).  Returns [(i0, i1)] (possibly empty ranges are dropped like empty caches would be)."""
    gen = get_segment_generator(shots)
    gen.send(None)
    out, start = [], 0
    for i, t in enumerate(times):
        if gen.send(t):
            out.append((start, i))
            start = i
    out.append((start, len(times)))
    return out


def resident_source(job, frames, times, shots, every, resize=None):
    """shots of a video whose frames the caller holds (DeviceFrames / arrays, all of them, for the whole run)"""
    for i0, i1 in split_into_shots(times, shots):
        yield ShotInput(job, i0, [(times[i], frames[i]) for i in range(i0, i1)], [(i % every == 0) for i in range(i0, i1)], resize=resize)
    yield JobEnd(job)


class StreamSource(object):
    """Reads `video` (an iterable of (t, frame): numpy uint8 [H, W, 3] or DeviceFrame) in a thread of its own, cuts it into shots by the
    reference's flush rule and hands complete shots to the engine through a bounded queue.  numpy frames reach HBM through a pinned
    ingest ring (one asynchronous copy each on the copy stream; the reference's `Video.__iter__` + `np.fromstring`, video.py:368-406);
    the frames it staged are the engine's to release."""

    def __init__(self, ctx, jobs, depth=1, ring_depth=24):
        """jobs: [(job, video iterable, shots, every, resize or None)] read one after the other"""
        self.ctx, self.jobs = ctx, jobs
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self.ring_depth = int(ring_depth)
        self.error = None
        self.frames_read = 0
        self._stop = False
        self.th = threading.Thread(target=self._produce, name="pvface-ingest")
        self.th.start()

    def _produce(self):
        try:
            for job, video, shots, every, resize in self.jobs:
                seg = get_segment_generator(shots)
                seg.send(None)
                ring = None
                cache, flags, base, i, owned = [], [], 0, 0, False
                for t, frame in video:
                    if self._stop:
                        return
                    if seg.send(t):
                        self.q.put(ShotInput(job, base, cache, flags, owned=owned, resize=resize))
                        cache, flags, base = [], [], i
                    if isinstance(frame, np.ndarray):
                        if ring is None or (ring.h, ring.w) != frame.shape[:2]:
                            if ring is not None:
                                ring.close()
                            ring = self.ctx.ingest_ring(frame.shape[0], frame.shape[1], depth=self.ring_depth)
                        frame = ring.push(frame)
                        owned = True
                    cache.append((t, frame))
                    flags.append(i % every == 0)
                    i += 1
                    self.frames_read += 1
                self.q.put(ShotInput(job, base, cache, flags, owned=owned, resize=resize))
                self.q.put(JobEnd(job))
                if ring is not None:
                    ring.close()              # waits for the last uploads
        except BaseException as e:              # noqa: BLE001 -- re-raised in the consumer
            self.error = e
        finally:
            self.q.put(None)

    def __iter__(self):
        while True:
            try:
                item = self.q.get(timeout=0.1)
            except queue.Empty:
                if not self.th.is_alive() and self.q.empty():      # the producer is gone and so is its end mark (close() took it)
                    return
                continue
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            yield item

    def close(self):
        """stop reading (an error downstream, a consumer that went away) and wait for the reader thread"""
        self._stop = True
        while self.th.is_alive():
            try:
                release_shot_frames(self.q.get(timeout=0.05))      # a shot nobody will process: its staged frames go back to the pool
            except queue.Empty:
                pass
        self.th.join()
        while True:
            try:
                release_shot_frames(self.q.get_nowait())
            except queue.Empty:
                break
