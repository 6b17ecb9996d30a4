"""End-to-end detect -> track -> landmarks -> embed -> cluster over frames resident in HBM.

This is the in-memory equivalent of the reference's three stages run back to back
    pyannote-face.py track   (scripts/pyannote-face.py:239-269)   -> track.txt
    pyannote-face.py extract (:271-314)                            -> landmarks.txt, embedding.txt
    FaceClustering().model.preprocess(embedding.txt) + __call__   (face/clustering.py:59-82,130-134)
with the inter-stage text formats applied in memory (3-decimal boxes + int() truncation, 5-decimal embeddings,
`getFaceGenerator`'s time synchronisation incl. its habit of never emitting the last timestamp group), so the integer
results equal what the file-based flow produces.  The video is staged once; the reference decodes it twice.
"""
import time as _time
import os
import numpy as np
from . import formats
from .face_tracking import FaceTracking
from .tracking_by_detection import get_segment_generator, HipTrackers
from .clustering import FaceClustering

# CLI defaults of `pyannote-face.py track` (scripts/pyannote-face.py:112-114) -- they differ from the API defaults
CLI_MIN_OVERLAP_RATIO = 0.5
CLI_MIN_CONFIDENCE = 10.
CLI_MAX_GAP = 1.


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


def faces_per_frame(rows, frame_times, frame_width, frame_height, drop_last=True):
    """Replays getFaceGenerator (pyannote-face.py:121-175): rows = [(T, id, box_norm_f32, status)] sorted by T.
    Returns [(frame index, T, [(id, (l,t,r,b) int)])] for the frames that receive faces."""
    out = []
    k, n = 0, len(rows)
    groups = []   # (T, [(id, box)]) in order; the LAST group is never emitted by the reference's generator
    while k < n:
        T = rows[k][0]
        g = []
        while k < n and rows[k][0] == T:
            _, ident, box, _ = rows[k]
            g.append((ident, formats.denormalise(box, frame_width, frame_height)))
            k += 1
        groups.append((T, g))
    if drop_last:
        groups = groups[:-1]   # a shard that is not the end of the video keeps its last group (see dist.py)
    gi = 0
    for fi, t in enumerate(frame_times):
        if gi >= len(groups):
            break
        T, g = groups[gi]
        if T > t:
            continue
        out.append((fi, T, g))
        gi += 1
    return out


class ExtractStream(object):
    """`extract` (pyannote-face.py:121-175, 425-466) fed shot by shot while later shots are still being detected.

    The reference reads the finished track file and walks frames and timestamp groups in step (faces_per_frame above).  Shots
    are disjoint in time and arrive in order, so the same walk can be resumed whenever a shot's tracks exist: groups are
    appended to a queue, the frame pointer only moves while a group is available, and the newest group is held back until a
    later one arrives because the reference's generator never yields the last group of the file.  The faces that become
    available are aligned and embedded immediately."""

    def __init__(self, ctx, frames, frame_times, frame_width, frame_height):
        self.ctx, self.frames, self.times = ctx, frames, frame_times
        self.w, self.h = frame_width, frame_height
        self.tracks, self.rows = [], []
        self.file_T, self.file_id = [], []    # the track file's (T, track) column in file order (decides the row order of the outputs)
        self.groups, self.gi, self.fi = [], 0, 0
        self.face_boxes, self.face_T, self.face_id = [], [], []
        self.pts, self.emb = [], []
        self.emitted = []     # (frame index, T) of every group handed on, in order

    def _emit(self, available):
        """faces of the groups that may be handed on now: (frames, boxes)"""
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
        return face_frames, boxes

    def compute(self, work):
        """GPU part: landmarks + embeddings of one batch of faces returned by prepare(); batches must arrive in order"""
        face_frames, boxes = work
        if boxes:
            pts = self.ctx.landmarks(face_frames, boxes)
            self.pts.append(pts)
            self.emb.append(self.ctx.embed(face_frames, pts))

    def prepare(self, tracks):
        """host part for the normalised tracks of the next shot (in shot order): the track-file rows, their timestamp groups,
        and the faces that can be extracted now"""
        base = len(self.tracks)
        # the track file's rows: time and box with 3 decimals ('%.3f'), the box then parsed as float32 (pyannote-face.py:125-127).
        # round(v, 3) is the correctly rounded 3-decimal value, i.e. float('%.3f' % v); the float32 parse is one array cast.
        flat = [round(v, 3) for track in tracks for _, box, _ in track for v in box]
        q32 = np.asarray(flat, np.float64).astype(np.float32).astype(np.float64).reshape(-1, 4).tolist() if flat else []
        rows, i = [], 0
        for k, track in enumerate(tracks):
            for t, _, status in track:
                rows.append((round(t, 3), base + k, tuple(q32[i]), status))
                i += 1
        self.file_T.extend(r[0] for r in rows)
        self.file_id.extend(r[1] for r in rows)
        rows.sort(key=lambda r: r[0])
        self.tracks.extend(tracks)
        self.rows.extend(rows)
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
        order = formats.pandas_sort_order(self.file_T)
        by_key = {(r[0], r[1]): r for r in self.rows}
        self.rows = [by_key[(self.file_T[i], self.file_id[i])] for i in order]
        self._planned = True

    def finish(self, drop_last=True, reorder=True):
        """reorder: put the faces of one timestamp into the order the reference's `extract` writes them (formats.file_order).
        A shard of a longer video leaves that to the step that sees the whole track table (dist.gather_rows)."""
        if not getattr(self, "_planned", False):
            self.plan_finish(drop_last, reorder)
        self.compute(self._final_work)
        pts = np.concatenate(self.pts) if self.pts else np.zeros((0, 68, 2), np.int32)
        emb = np.concatenate(self.emb) if self.emb else np.zeros((0, 128), np.float32)
        if self._perm is not None:
            pts, emb = pts[self._perm], emb[self._perm]
        return pts, emb


def detections_as_lists(n_frames, raw):
    """[[(l, t, r, b) Python ints]] per frame from the arrays of Context.detect_many(arrays=True): raw = (boxes, counts, frame indices)"""
    dets = [[] for _ in range(n_frames)]
    if raw is not None:
        out, cnt, idx = raw
        rows, cnt = out.tolist(), cnt.tolist()
        for j, i in enumerate(idx):
            dets[i] = [tuple(b) for b in rows[j][:cnt[j]]]
    return dets


class _LaneBackend(object):
    """what the lanes of the tracking thread see of the tracker context while the GPU thread owns it: on-demand updates
    take the context lock, killed trackers are only queued (the GPU thread destroys them between its batches)"""

    def __init__(self, backend, lock, dead):
        self.backend, self.lock, self.dead = backend, lock, dead

    def update_many(self, handles, frames):
        with self.lock:
            return self.backend.update_many(handles, frames)

    def commit_many(self, handles, frames):
        with self.lock:
            self.backend.commit_many(handles, frames)

    def start_many(self, frames, boxes):
        with self.lock:
            return self.backend.start_many(frames, boxes)

    def release(self, handle):
        self.dead.append(handle)


class FacePipeline(object):
    def __init__(self, ctx, landmarks, embedding, detect_min_size=0.0, detect_every=0.0,
                 track_min_overlap_ratio=CLI_MIN_OVERLAP_RATIO, track_min_confidence=CLI_MIN_CONFIDENCE,
                 track_max_gap=CLI_MAX_GAP, threshold=0.6, detect_batch_size=8, overlap=True):
        self.ctx = ctx
        # overlap=True: ONE host thread feeds the GPU with large batches in a fixed order -- detect shot k, bulk tracker work of
        # shot k, align + embed the faces of shot k-1 -- while the caller's thread runs the host state machine (association,
        # graph, merging) of shot k one step behind.  Every kernel runs uncontended on one stream; the Python work hides
        # behind the GPU work.  (Several streams with a detector thread running ahead were measured slower: the small
        # tracker batches they need fill the GPU badly and the threads fight over the interpreter lock.)
        self.overlap = overlap
        self.detect_batch_size = detect_batch_size
        ctx.load_shape_predictor(landmarks)
        ctx.load_embedder(embedding)
        self.tracking = FaceTracking(detect_min_size=detect_min_size, detect_every=detect_every,
                                     track_min_confidence=track_min_confidence,
                                     track_min_overlap_ratio=track_min_overlap_ratio,
                                     track_max_gap=track_max_gap, ctx=ctx, detect_batch_size=detect_batch_size)
        self.clustering = FaceClustering(threshold=threshold, ctx=ctx)
        self.detect_every = detect_every
        self.detect_min_size = detect_min_size

    def _run_pipelined(self, shot_inputs, backend, ex, normalize, mark, before_join=None):
        """GPU thread: detect(k), speculate(k), extract(k-1) ...; this thread: lanes + merging of shot k as soon as its detections
        and bulk tracker results exist.  ctypes releases the GIL inside every library call."""
        import threading
        import queue
        n = len(shot_inputs)
        ready, done = queue.Queue(), queue.Queue()
        lock = threading.Lock()
        dead = []
        bs = max(1, int(self.detect_batch_size))
        ctx = self.ctx
        trace = [(_time.perf_counter(), "begin")] if os.environ.get("PVF_TRACE") else None

        def note(*ev):
            if trace is not None:
                trace.append((_time.perf_counter(),) + ev)

        def release_dead():
            batch = []
            while dead:
                batch.append(dead.pop())
            if batch:
                backend.release_many(batch)

        def gpu_thread():
            extracted = 0
            try:
                for k, (cache, flags) in enumerate(shot_inputs):
                    idx = [i for i, f in enumerate(flags) if f]
                    counts = np.zeros(len(cache), np.int64)
                    boxes = np.zeros((0, 4), np.float64)
                    raw = None
                    note("detect begin", k)
                    if idx:
                        with lock:
                            out, _, cnt = ctx.detect_many([cache[i][1] for i in idx], bs, 1, arrays=True)
                        # the boxes go back to the GPU (tracker starts) as an array; the tracking thread turns them into the Python
                        # tuples its state machine works on while the GPU is busy with those starts
                        counts[idx] = cnt
                        boxes = out[np.arange(out.shape[1])[None, :] < cnt[:, None]].astype(np.float64)
                        raw = (out, cnt, idx)
                    note("detected", k)

                    def drain():
                        nonlocal extracted
                        while extracted < n:
                            try:
                                work = done.get_nowait()
                            except queue.Empty:
                                break
                            note("extract begin", extracted)
                            with lock:
                                ex.compute(work)
                            extracted += 1
                            note("extracted", extracted - 1)

                    # faces of the shots the tracking thread has finished meanwhile (it is idle now, so the host side of these
                    # calls does not compete with its state machine for the interpreter; measured better than after speculate).
                    # Towards the end the order changes: the faces of the last TWO finished shots are held back until the last
                    # shot's bulk tracker work is queued, so that its state machine (17-21 ms on the host, plus its on-demand
                    # tracker calls) runs beside ~34 ms of embedding instead of leaving the GPU idle at the very end.
                    if k < n - 2:
                        drain()
                    note("speculate begin", k)
                    with lock:
                        release_dead()
                        if hasattr(backend, "speculate_pair"):
                            plan_f, plan_b = backend.speculate_pair(cache, None, counts=counts, boxes=boxes)
                        else:
                            det_at = {t: d for (t, _), d in zip(cache, detections_as_lists(len(cache), raw))}
                            plan_f = backend.speculate(cache, det_at)
                            plan_b = backend.speculate(list(reversed(cache)), det_at)
                    note("speculated", k)
                    ready.put((k, raw, (plan_f, plan_b)))
                    if k == n - 1:
                        drain()
                while extracted < n:
                    work = done.get()
                    if work is None:
                        return
                    note("extract begin", extracted)
                    with lock:
                        ex.compute(work)
                    extracted += 1
                    note("extracted", extracted - 1)
                with lock:
                    release_dead()
            except BaseException as e:
                ready.put(e)

        import sys
        old_interval = sys.getswitchinterval()
        # the GPU thread re-takes the interpreter lock after every library call; with the default 5 ms switch interval each of
        # those hand-overs can stall the GPU queue for milliseconds while this thread runs the tracking state machine
        sys.setswitchinterval(1e-4)
        th = threading.Thread(target=gpu_thread, name="pvface-gpu")
        th.start()
        lane_backend = _LaneBackend(backend, lock, dead)
        ok = False
        try:
            for k, (cache, flags) in enumerate(shot_inputs):
                item = ready.get()
                if isinstance(item, BaseException):
                    raise item
                _, raw, plans = item
                note("host begin", k)
                dets = detections_as_lists(len(cache), raw)
                job = self.tracking.begin_shot(cache, flags, dets, lane_backend, plans)
                self.tracking._run_lanes(job["lanes"], lane_backend)
                note("lanes done", k)
                tracks = self.tracking.finish_shot(job)
                note("tracked", k)
                if k == n - 1:
                    mark["tracked"] = _time.perf_counter()
                done.put(ex.prepare(normalize(tracks)))
                note("prepared", k)
            if before_join is not None:
                before_join()                 # host work that only needs the tracks: runs while the GPU thread embeds the last faces
                note("planned")
            ok = True
        finally:
            if not ok:
                done.put(None)
            th.join()
            sys.setswitchinterval(old_interval)
        if not ready.empty():
            item = ready.get()
            if isinstance(item, BaseException):
                raise item
        if trace is not None:
            import json
            note("finish")
            with open(os.environ["PVF_TRACE"], "w") as f:
                json.dump(trace, f)

    def run(self, frames, times, frame_rate, shots, timings=None, cluster=True, last_shard=True, reorder=True):
        """frames: list of DeviceFrame (or numpy arrays), one size; times: their timestamps; shots: [(start, end)].
        Returns dict(tracks, track_rows, faces, landmarks, embeddings, labels)."""
        tm = timings if timings is not None else {}
        t0 = _time.perf_counter()
        h, w = frames[0].shape[0], frames[0].shape[1]
        every = int(self.detect_every * frame_rate) if self.detect_every > 0.0 else 1
        every = max(every, 1)
        ranges = split_into_shots(times, shots)
        # --min-size: detection and tracking run on frames scaled down so that the smallest face wanted is ~36 px tall; boxes are
        # normalised by that size (tracking.py:389-400,414) and `extract` works on the native frames (pyannote-face.py:275-277)
        tw, th, track_frames = w, h, frames
        if self.detect_min_size > 0.0:
            ratio = min(1.0, self.tracking.detect_smallest / (self.detect_min_size * h))
            tw, th = int(w * ratio), int(h * ratio)
            if (tw, th) != (w, h):
                track_frames = [self.ctx.resize(f, tw, th) for f in frames]
        shot_inputs = []
        for i0, i1 in ranges:
            cache = [(times[i], track_frames[i]) for i in range(i0, i1)]
            flags = [(i % every == 0) for i in range(i0, i1)]
            shot_inputs.append((cache, flags))
        backend = HipTrackers(self.ctx)
        ex = ExtractStream(self.ctx, frames, times, w, h)
        mark = {}

        def normalize(shot_tracks):
            return [self.tracking._normalize_track(tr, tw, th) for tr in shot_tracks]

        if not self.overlap:
            for k, shot_tracks in enumerate(self.tracking.process_shots(shot_inputs, backend)):
                if k == len(shot_inputs) - 1:
                    mark["tracked"] = _time.perf_counter()
                ex.feed(normalize(shot_tracks))
        else:
            import gc
            was_enabled = gc.isenabled()
            gc.disable()      # a full collection in the middle of a shot stalls both threads for tens of milliseconds
            try:
                self._run_pipelined(shot_inputs, backend, ex, normalize, mark,
                                    before_join=lambda: ex.plan_finish(drop_last=last_shard, reorder=reorder))
            finally:
                if was_enabled:
                    gc.enable()
        pts, emb = ex.finish(drop_last=last_shard, reorder=reorder)
        tracks, rows = ex.tracks, ex.rows
        face_boxes, face_T, face_id = ex.face_boxes, ex.face_T, ex.face_id
        t1 = mark.get("tracked", _time.perf_counter())
        tm["track_s"] = t1 - t0                                  # until the last shot's tracks exist (earlier shots already extracted)
        tm["extract_s"] = _time.perf_counter() - t1              # what is left of extraction after that
        t2 = _time.perf_counter()
        face_T = np.asarray(face_T, np.float64)
        face_id = np.asarray(face_id, np.int64)
        Xq = np.round(emb.astype(np.float64), 5) if len(emb) else np.zeros((0, 128))
        # np.round(.,5) of the float64 value == parsing '%.5f' text for these magnitudes; formats.quantise_embedding is the literal form
        labels = {}
        if len(face_T) and cluster:
            starting_point, data = self.clustering.model.preprocess((face_T, face_id, Xq))
            result = self.clustering(starting_point, features=data)
            labels = {int(track): int(label) for _, track, label in result.itertracks(yield_label=True)}
        tm["cluster_s"] = _time.perf_counter() - t2
        tm["total_s"] = _time.perf_counter() - t0
        return {"tracks": tracks, "track_rows": rows, "face_T": face_T, "face_id": face_id, "face_boxes": face_boxes,
                "landmarks": pts, "embeddings": emb, "X": Xq, "labels": labels, "shot_ranges": ranges,
                "file_T": np.asarray(ex.file_T, np.float64), "file_id": np.asarray(ex.file_id, np.int64)}


def _noop():
    pass


def detector_geometry(height, width, upsample=1, cell=8, frows=10, fcols=10, min_w=64, min_h=64):
    """Level schedule of the HOG scanner for one frame size (same integer rules as csrc/detect.hip and the oracle).
    Returns [(img_h, img_w, feat_h, feat_w, positions)] -- used for the algorithmic work in bench.py / DESIGN.md."""
    import math
    h, w = height, width
    for _ in range(upsample):
        w, h = int(math.floor(((w - 1) + 1.25) * 2.0 + 0.5)) + 1, int(math.floor(((h - 1) + 0.75) * 2.0 + 0.5)) + 1
    rnd = lambda v: int(math.floor(v + 0.5))
    r = [0, 0, w - 1, h - 1]
    levels = 0
    while True:
        r = [rnd((v - 0.3) * (5.0 / 6.0) + 0.3) for v in r]
        levels += 1
        if not ((r[2] - r[0] + 1) >= min_w and (r[3] - r[1] + 1) >= min_h and levels < 1000):
            break
    out = []
    for l in range(levels):
        if l > 0:
            h, w = (5 * h) // 6, (5 * w) // 6
        cr, cc = int(h / float(cell) + 0.5), int(w / float(cell) + 0.5)
        fh, fw = cr - 2 + frows - 1, cc - 2 + fcols - 1
        pos = max(fh - (frows - 1), 0) * max(fw - (fcols - 1), 0) if (fh >= frows and fw >= fcols) else 0
        out.append((h, w, fh, fw, pos))
    return out
