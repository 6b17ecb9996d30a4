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
from . import _lib
from .face_tracking import FaceTracking
from .tracking_by_detection import HipTrackers
from .clustering import FaceClustering
from . import engine as _engine
from .engine import ExtractStream, split_into_shots, detections_as_lists      # noqa: F401  (part of this module's surface)

# CLI defaults of `pyannote-face.py track` (scripts/pyannote-face.py:112-114) -- they differ from the API defaults
CLI_MIN_OVERLAP_RATIO = 0.5
CLI_MIN_CONFIDENCE = 10.
CLI_MAX_GAP = 1.


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




class FacePipeline(object):
    def __init__(self, ctx, landmarks, embedding, detect_min_size=0.0, detect_every=0.0,
                 track_min_overlap_ratio=CLI_MIN_OVERLAP_RATIO, track_min_confidence=CLI_MIN_CONFIDENCE,
                 track_max_gap=CLI_MAX_GAP, threshold=0.6, detect_batch_size=8, overlap=True,
                 speculate_limit=8192, speculate_window=4096):
        self.ctx = ctx
        # overlap=True: ONE host thread feeds the GPU with large batches in a fixed order -- detect shot k, bulk tracker work of
        # shot k, align + embed the faces of shot k-1 -- while the caller's thread runs the host state machine (association,
        # graph, merging) of shot k one step behind.  Every kernel runs uncontended on one stream; the Python work hides
        # behind the GPU work.  (Several streams with a detector thread running ahead were measured slower: the small
        # tracker batches they need fill the GPU badly and the threads fight over the interpreter lock.)
        self.overlap = overlap
        self.detect_batch_size = detect_batch_size
        if landmarks is not None:
            ctx.load_shape_predictor(landmarks)
        if embedding is not None:
            ctx.load_embedder(embedding)
        self.can_extract = landmarks is not None and embedding is not None
        self.tracking = FaceTracking(detect_min_size=detect_min_size, detect_every=detect_every,
                                     track_min_confidence=track_min_confidence,
                                     track_min_overlap_ratio=track_min_overlap_ratio,
                                     track_max_gap=track_max_gap, ctx=ctx, detect_batch_size=detect_batch_size)
        self.clustering = FaceClustering(threshold=threshold, ctx=ctx)
        self.detect_every = detect_every
        self.return_table = True       # results carry "X", the float64 table of the clustering (a host pass over all descriptors)
        self.detect_min_size = detect_min_size
        # trackers a shot may hold at once before its bulk starts are windowed (2.39 MB of filters each): engine.WindowedPlan
        self.speculate_limit, self.speculate_window = speculate_limit, speculate_window
        self.shot_group = 8              # with --every: shots whose passes share their frame-by-frame tracker calls (engine.Engine.group)
        self.farm_extract_min = 3072        # run_many: faces that wait together before an extraction call is made (at most 4096 go into one)
        self.last_engine = None

    # ---- geometry of one video -------------------------------------------------------------------------------------------------
    def _every(self, frame_rate):
        every = int(self.detect_every * frame_rate) if self.detect_every > 0.0 else 1
        return max(every, 1)

    def _detection_size(self, w, h):
        """--min-size: detection and tracking run on frames scaled down so that the smallest face wanted is ~36 px tall; boxes are
        normalised by that size (tracking.py:389-400,414) and `extract` works on the native frames (pyannote-face.py:275-277)"""
        if self.detect_min_size > 0.0:
            ratio = min(1.0, self.tracking.detect_smallest / (self.detect_min_size * h))
            return int(w * ratio), int(h * ratio)
        return w, h

    def _engine(self, extract_min=0):
        e = _engine.Engine(self.ctx, self.tracking, detect_batch_size=self.detect_batch_size, overlap=self.overlap,
                           speculate_limit=self.speculate_limit, speculate_window=self.speculate_window,
                           group=(self.shot_group if self.detect_every > 0.0 else 1), extract_min=extract_min)
        self.last_engine = e
        return e

    def _result(self, job, tm, cluster, t0):
        """a finished job -> the result dictionary (`extract`'s arrays in file order, clustering on request)"""
        if job.ex is None:
            if tm is not None:
                tm["track_s"] = tm["total_s"] = _time.perf_counter() - t0
            return {"tracks": job.tracks, "shot_ranges": job.shot_ranges, "labels": {}}
        ex = job.ex
        pts, emb = ex.finish(drop_last=job.last_shard, reorder=job.reorder, computed=True)
        t2 = _time.perf_counter()
        face_T = np.asarray(ex.face_T, np.float64)
        face_id = np.asarray(ex.face_id, np.int64)
        labels = {}
        if len(face_T) and cluster:
            # FaceClustering()(*model.preprocess(embedding.txt)) of the reference (clustering.py:130-134) on the rows in memory: the
            # float32 descriptors go up once (4 bytes per value); the table the reference reads back from the '%.5f' text --
            # np.round(float64(x), 5), rows in (track, time) order -- is made on the device, followed by the pair means and the agglomeration
            labels = self.clustering.cluster_rows(face_T, face_id, emb)
        # "X": that table on the host (np.round(x, 5) of the float64 value == parsing the '%.5f' text for these magnitudes;
        # formats.quantise_embedding is the literal form), for callers that want it; the clustering above does not need it
        Xq = (_lib.round_rows(emb, 5) if len(emb) else np.zeros((0, 128))) if self.return_table else None
        if tm is not None:
            tm["cluster_s"] = _time.perf_counter() - t2
            t1 = job.t_tracked if job.t_tracked is not None else t2
            tm["track_s"] = t1 - t0                              # until the last shot's tracks exist (earlier shots already extracted)
            tm["extract_s"] = t2 - t1                            # what is left of extraction after that
            tm["total_s"] = _time.perf_counter() - t0
        return {"tracks": ex.tracks, "track_rows": ex.rows, "face_T": face_T, "face_id": face_id, "face_boxes": ex.face_boxes,
                "landmarks": pts, "embeddings": emb, "X": Xq, "labels": labels, "shot_ranges": job.shot_ranges,
                "file_T": np.asarray(ex.file_T, np.float64), "file_id": np.asarray(ex.file_id, np.int64)}

    # ---- frames the caller holds ----------------------------------------------------------------------------------------------------
    def run(self, frames, times, frame_rate, shots, timings=None, cluster=True, last_shard=True, reorder=True, extract=True):
        """frames: list of DeviceFrame (or numpy arrays), one size, all resident for the whole run; times: their timestamps;
        shots: [(start, end)].  Returns dict(tracks, track_rows, faces, landmarks, embeddings, labels)."""
        t0 = _time.perf_counter()
        h, w = frames[0].shape[0], frames[0].shape[1]
        if any(isinstance(f, np.ndarray) for f in frames):
            frames = [self.ctx.upload(f) if isinstance(f, np.ndarray) else f for f in frames]     # staged once, up front, for the whole run
        tw, th = self._detection_size(w, h)
        job = _engine.VideoJob(self.ctx, w, h, tw, th, frames=frames, times=times, extract=extract and self.can_extract,
                               last_shard=last_shard, reorder=reorder)
        ranges = split_into_shots(times, shots)
        source = _engine.resident_source(job, frames, times, shots, self._every(frame_rate), resize=(tw, th) if (tw, th) != (w, h) else None)
        self._engine().run(source, HipTrackers(self.ctx), n_shots=len(ranges))
        return self._result(job, timings, cluster, t0)

    # ---- frames that stream in ------------------------------------------------------------------------------------------------------
    def run_stream(self, video, shots, frame_rate=None, size=None, timings=None, cluster=True, extract=True, on_tracks=None,
                   last_shard=True, reorder=True, queue_depth=1, ring_depth=24):
        """`video`: an iterable of (t, frame) -- numpy uint8 [H, W, 3] or DeviceFrame -- read ONCE, in a thread of its own, shot by shot
        (the reference decodes the video twice: pyannote-face.py:261 and :287); `shots`: segments (with .end) or (start, end) pairs.
        Frames the source hands over as numpy arrays go through the pinned ingest ring and are released as soon as `extract` has passed
        them, so the run holds the shots in flight, not the video.  on_tracks(tracks): called with every shot's normalised tracks, in
        order (the `track` verb writes its file from it).  Same result dictionary as run()."""
        t0 = _time.perf_counter()
        frame_rate = float(frame_rate if frame_rate is not None else video.frame_rate)
        w, h = size if size is not None else video.size
        tw, th = self._detection_size(w, h)
        job = _engine.VideoJob(self.ctx, w, h, tw, th, extract=extract and self.can_extract, last_shard=last_shard, reorder=reorder,
                               on_tracks=on_tracks)
        src = _engine.StreamSource(self.ctx, [(job, video, shots, self._every(frame_rate), (tw, th) if (tw, th) != (w, h) else None)],
                                   depth=queue_depth, ring_depth=ring_depth)
        try:
            self._engine().run(src, HipTrackers(self.ctx))
        finally:
            src.close()
        res = self._result(job, timings, cluster, t0)
        res["frames"] = src.frames_read
        res["peak_frames_resident"] = getattr(job.store, "peak", None)
        return res

    # ---- many independent videos (BASELINE.json configs[3]: clips farmed to a GPU, no collective) ----------------------------------
    def run_many(self, clips, cluster=True, extract=True, on_result=None):
        """clips: [dict(frames=[...], times=[...], frame_rate=, shots=[...])] (resident) or [dict(video=iterable, shots=[...])] (streamed;
        `video` needs .size and .frame_rate).  Every clip is a job of its own -- own track numbering, own `extract` walk, own clustering
        (face/clustering.py:130-134 per clip) -- but all share ONE engine run, so the detector of clip i + 1 runs while the state
        machine of clip i is busy on the host and the clustering of clip i - 1 runs beside both.  Returns the result dictionaries in
        clip order; on_result(index, result) is called as soon as a clip is complete."""
        t0 = _time.perf_counter()
        jobs, resident, streamed = [], [], []
        n_shots = 0
        for k, c in enumerate(clips):
            if "frames" in c:
                frames, times = c["frames"], c["times"]
                h, w = frames[0].shape[0], frames[0].shape[1]
                tw, th = self._detection_size(w, h)
                job = _engine.VideoJob(self.ctx, w, h, tw, th, frames=frames, times=times, extract=extract and self.can_extract, key=k)
                resident.append((job, frames, times, c["shots"], self._every(c["frame_rate"]), (tw, th) if (tw, th) != (w, h) else None))
                n_shots += len(split_into_shots(times, c["shots"]))
            else:
                video = c["video"]
                w, h = video.size
                tw, th = self._detection_size(w, h)
                job = _engine.VideoJob(self.ctx, w, h, tw, th, extract=extract and self.can_extract, key=k)
                streamed.append((job, video, c["shots"], self._every(float(video.frame_rate)), (tw, th) if (tw, th) != (w, h) else None))
            jobs.append(job)
        if resident and streamed:
            raise ValueError("run_many: resident and streamed clips cannot be mixed in one call")
        results = [None] * len(jobs)

        def complete(job):
            results[job.key] = self._result(job, None, cluster, t0)
            if on_result is not None:
                on_result(job.key, results[job.key])

        src = None
        if streamed:
            src = source = _engine.StreamSource(self.ctx, streamed)
        else:
            def source_gen():
                for args in resident:
                    for item in _engine.resident_source(*args):
                        yield item
            source = source_gen()
        try:
            # many short videos: their faces are extracted a few thousand at a time, whichever clips they belong to (engine.compute_many)
            self._engine(extract_min=(self.farm_extract_min if len(jobs) > 1 else 0)).run(
                source, HipTrackers(self.ctx), n_shots=(n_shots if not streamed else None), on_job_final=complete)
        finally:
            if src is not None:
                src.close()
        return results


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
