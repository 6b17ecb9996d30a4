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
            g.append((ident, (int(box[0] * frame_width), int(box[1] * frame_height),
                              int(box[2] * frame_width), int(box[3] * frame_height))))
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


class FacePipeline(object):
    def __init__(self, ctx, landmarks, embedding, detect_min_size=0.0, detect_every=0.0,
                 track_min_overlap_ratio=CLI_MIN_OVERLAP_RATIO, track_min_confidence=CLI_MIN_CONFIDENCE,
                 track_max_gap=CLI_MAX_GAP, threshold=0.6, detect_batch_size=8, overlap=True):
        self.ctx = ctx
        # The detector is throughput-bound (big kernels), the trackers are latency-bound (many tiny dependent launches):
        # a second context (own HIP stream + scratch) lets a host thread run detection ahead, shot by shot, while the
        # main thread tracks the previous shot.  Both streams share the GPU; results are identical to the serial order.
        self.det_ctx = None
        if overlap:
            from .runtime import Context
            self.det_ctx = Context(device=ctx.device, priority=-1)
        self.detect_batch_size = detect_batch_size
        ctx.load_shape_predictor(landmarks)
        ctx.load_embedder(embedding)
        self.tracking = FaceTracking(detect_min_size=detect_min_size, detect_every=detect_every,
                                     track_min_confidence=track_min_confidence,
                                     track_min_overlap_ratio=track_min_overlap_ratio,
                                     track_max_gap=track_max_gap, ctx=ctx, detect_batch_size=detect_batch_size)
        self.clustering = FaceClustering(threshold=threshold, ctx=ctx)
        self.detect_every = detect_every

    def _track_overlapped(self, shot_inputs, backend):
        """detector thread (second context) runs ahead shot by shot; the caller's thread tracks shot k as soon as its
        detections exist.  ctypes releases the GIL inside every library call."""
        import threading
        dctx = self.det_ctx
        n = len(shot_inputs)
        dets = [None] * n
        ready = [threading.Event() for _ in range(n)]
        err = []
        bs = max(1, int(self.detect_batch_size))

        def worker():
            try:
                for k, (cache, flags) in enumerate(shot_inputs):
                    out = [[] for _ in cache]
                    idx = [i for i, f in enumerate(flags) if f]
                    shared = {i: dctx.share(cache[i][1]) if hasattr(cache[i][1], "handle") else cache[i][1] for i in idx}
                    for o in range(0, len(idx), bs):
                        chunk = idx[o:o + bs]
                        res = dctx.detect_batch([shared[i] for i in chunk], 1)
                        for i, (boxes, _) in zip(chunk, res):
                            out[i] = [tuple(b) for b in boxes]
                    dets[k] = out
                    ready[k].set()
            except BaseException as e:   # surface in the caller's thread
                err.append(e)
                for ev in ready:
                    ev.set()

        th = threading.Thread(target=worker, name="pvface-detector")
        th.start()
        from .tracking_by_detection import LaneScheduler
        sched = LaneScheduler(backend)
        jobs = [None] * n
        nxt = 0
        try:
            while nxt < n or len(sched):
                # admit every shot whose detections exist; block only when there is nothing to track
                while nxt < n and (ready[nxt].is_set() or len(sched) == 0):
                    ready[nxt].wait()
                    if err:
                        raise err[0]
                    cache, flags = shot_inputs[nxt]
                    jobs[nxt] = self.tracking.begin_shot(cache, flags, dets[nxt], backend)
                    for lane in jobs[nxt]["lanes"]:
                        sched.add(lane)
                    nxt += 1
                if len(sched):
                    sched.round()
        finally:
            th.join()
        if err:
            raise err[0]
        return [self.tracking.finish_shot(j) for j in jobs]

    def run(self, frames, times, frame_rate, shots, timings=None, cluster=True, last_shard=True):
        """frames: list of DeviceFrame (or numpy arrays), one size; times: their timestamps; shots: [(start, end)].
        Returns dict(tracks, track_rows, faces, landmarks, embeddings, labels)."""
        tm = timings if timings is not None else {}
        t0 = _time.perf_counter()
        h, w = frames[0].shape[0], frames[0].shape[1]
        every = int(self.detect_every * frame_rate) if self.detect_every > 0.0 else 1
        every = max(every, 1)
        ranges = split_into_shots(times, shots)
        shot_inputs = []
        for i0, i1 in ranges:
            cache = [(times[i], frames[i]) for i in range(i0, i1)]
            flags = [(i % every == 0) for i in range(i0, i1)]
            shot_inputs.append((cache, flags))
        backend = HipTrackers(self.ctx)
        if self.det_ctx is None:
            per_shot = self.tracking.process_shots(shot_inputs, backend)
        else:
            per_shot = self._track_overlapped(shot_inputs, backend)
        tracks = [self.tracking._normalize_track(tr, w, h) for shot in per_shot for tr in shot]
        tm["track_s"] = _time.perf_counter() - t0
        t1 = _time.perf_counter()
        # track.txt in memory, then extract's view of it
        rows = []
        for identifier, track in enumerate(tracks):
            for t, box, status in track:
                rows.append((formats.quantise_time(t), identifier, tuple(np.float32("%.3f" % v) for v in box), status))
        rows.sort(key=lambda r: r[0])
        per_frame = faces_per_frame(rows, times, w, h, drop_last=last_shard)
        face_frames, face_boxes, face_T, face_id = [], [], [], []
        for fi, T, g in per_frame:
            for ident, box in g:
                face_frames.append(frames[fi]); face_boxes.append(box); face_T.append(T); face_id.append(ident)
        pts = self.ctx.landmarks(face_frames, face_boxes)
        emb = self.ctx.embed(face_frames, pts)
        tm["extract_s"] = _time.perf_counter() - t1
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
                "landmarks": pts, "embeddings": emb, "X": Xq, "labels": labels, "shot_ranges": ranges}


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
