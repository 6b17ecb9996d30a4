"""`pyannote-face` verbs on the HIP path (reference scripts/pyannote-face.py:29-89 usage, :239-314 bodies, :415-455 dispatch).

    python -m pyannote_video_amd track   [options] <video> <shot.json> <tracking>
    python -m pyannote_video_amd extract [options] <video> <tracking> <landmark_model> <embedding_model> <landmarks> <embeddings>
    python -m pyannote_video_amd cluster [options] <embeddings> <labels>
    python -m pyannote_video_amd process [options] <video> <shot.json> <landmark_model> <embedding_model> <tracking> <landmarks> <embeddings>

`track` and `extract` take the reference's arguments and options and write byte-compatible files (track.txt, landmarks.txt,
embedding.txt: formats.py).  `cluster` is the verb BASELINE.json's north_star names; the reference offers clustering through the API
only (face/clustering.py:130-134).  It writes `identifier label` lines, the file `demo --label` reads (pyannote-face.py:87,391-397).
`demo` (video rendering through moviepy) is out of scope.

<video>: decoding through ffmpeg (reference video.py:345-406) is not part of this path; the verbs read frames that are already
decoded -- a `.npy` file holding uint8 [N, H, W, 3] RGB frames (memory mapped; frame rate from --fps) -- or render the bench's
synthetic clip: `synthetic:<width>x<height>x<frames>[:shots[:faces[:seed]]]`.
<shot.json>: a JSON list of [start, end] seconds (what pyannote.core.json holds for a Timeline reduces to this for our purpose).
"""
import argparse
import json
import sys
import numpy as np
from . import formats

MIN_OVERLAP_RATIO = 0.5      # pyannote-face.py:112-114
MIN_CONFIDENCE = 10.
MAX_GAP = 1.


class NpyVideo(object):
    """decoded frames with the iteration contract of the reference's Video (video.py:411-464): yields (t, uint8 HxWx3 RGB)"""

    def __init__(self, path, frame_rate=25.0):
        self._frames = np.load(path, mmap_mode="r")
        if self._frames.dtype != np.uint8 or self._frames.ndim != 4 or self._frames.shape[3] != 3:
            raise IOError("%s: expected uint8 frames of shape [N, H, W, 3]" % path)
        self.frame_rate = float(frame_rate)
        self._size = (int(self._frames.shape[2]), int(self._frames.shape[1]))
        self._frame_size = self._size
        self.duration = len(self._frames) / self.frame_rate
        self.step, self.start, self.end = 1.0 / self.frame_rate, 0.0, self.duration      # what structure.Shot reads (video.py:186-190)

    @property
    def size(self):
        return self._size

    @property
    def frame_size(self):
        return self._frame_size

    @frame_size.setter
    def frame_size(self, value):
        self._frame_size = tuple(int(v) for v in value)     # applied by the consumer on the device (FaceTracking: detect_min_size)

    def __len__(self):
        return len(self._frames)

    def __iter__(self):
        for i in range(len(self._frames)):
            yield i / self.frame_rate, np.ascontiguousarray(self._frames[i])


def open_video(spec, frame_rate):
    if spec.startswith("synthetic:"):
        from . import synth
        parts = spec[len("synthetic:"):].split(":")
        w, h, n = (int(v) for v in parts[0].lower().split("x"))
        kw = {}
        for name, val in zip(("n_shots", "faces", "seed"), parts[1:]):
            kw[name] = int(val)
        return synth.SyntheticVideo(width=w, height=h, n_frames=n, frame_rate=frame_rate, **kw)
    return NpyVideo(spec, frame_rate)


class _Shot(object):
    def __init__(self, start, end):
        self.start, self.end = float(start), float(end)


def load_shots(path):
    with open(path) as f:
        data = json.load(f)
    if isinstance(data, dict):                      # pyannote.core.json Timeline: {"pyannote": "Timeline", "content": [{"start":..,"end":..}]}
        data = [(s["start"], s["end"]) for s in data.get("content", [])]
    return [_Shot(a, b) for a, b in data]


def auto_detect_batch(width, height):
    """frames whose pyramids and feature maps are resident together: 128 at 1080p (18 GB), fewer for larger frames"""
    return int(max(8, min(128, 128 * (1920 * 1080) // max(int(width) * int(height), 1))))


def _pipeline(video, ctx, landmark_model=None, embedding_model=None, **kw):
    from . import pipeline, runtime
    ctx = ctx or runtime.default_context()
    w, h = video.size
    return pipeline.FacePipeline(ctx, landmark_model, embedding_model, detect_batch_size=auto_detect_batch(w, h), **kw)


def track(video, shot, output, detect_min_size=0.0, detect_every=0.0, track_min_overlap_ratio=MIN_OVERLAP_RATIO,
          track_min_confidence=MIN_CONFIDENCE, track_max_gap=MAX_GAP, ctx=None):
    """Tracking by detection (pyannote-face.py:239-269): track.txt, one line per (track, frame), flushed per track.  The video is read
    once, in a thread of its own, into the pinned ingest ring; shots go through the streaming engine (engine.py): batched detection,
    bulk tracker work, the state machine one shot behind, frames released shot by shot."""
    pipe = _pipeline(video, ctx, detect_min_size=detect_min_size, detect_every=detect_every, track_min_overlap_ratio=track_min_overlap_ratio,
                     track_min_confidence=track_min_confidence, track_max_gap=track_max_gap)
    shots = load_shots(shot) if isinstance(shot, str) else shot
    state = {"next": 0}
    with open(output, 'w') as foutput:
        def write(tracks):
            for trk in tracks:
                for line in formats.track_lines(state["next"], trk):
                    foutput.write(line)
                foutput.flush()
                state["next"] += 1
        tm = {}
        res = pipe.run_stream(video, shots, extract=False, cluster=False, on_tracks=write, timings=tm)
        res["timings"] = tm
        return res


def extract(video, landmark_model, embedding_model, tracking, landmark_output, embedding_output, ctx=None, batch=2048, ahead=96):
    """Facial features (pyannote-face.py:271-314): landmarks.txt and embedding.txt for every face of the track file.  The faces
    are paired with frames by getFaceGenerator's rules (pipeline.faces_per_frame); a reader thread pushes the frames that carry faces
    through the pinned ingest ring (asynchronous uploads, at most `ahead` frames in flight) while this thread computes `batch` faces per
    library call and writes the lines in the reference's order."""
    import queue
    import threading
    from . import pipeline, runtime
    ctx = ctx or runtime.default_context()
    ctx.load_shape_predictor(landmark_model)
    ctx.load_embedder(embedding_model)
    frame_width, frame_height = video.frame_size
    rows = formats.read_tracks(tracking)
    frame_times = [t for t, _ in _times(video)]
    plan = pipeline.faces_per_frame(rows, frame_times, frame_width, frame_height)
    want = {}
    for fi, T, g in plan:
        want[fi] = (T, g)
    last_wanted = max(want) if want else -1
    q = queue.Queue(maxsize=max(2, int(ahead)))
    state = {"error": None}

    def reader():
        ring = None
        try:
            for fi, (t, rgb) in enumerate(video):
                if fi > last_wanted:
                    break
                if fi not in want:
                    continue
                owned = False
                if not isinstance(rgb, runtime.DeviceFrame):
                    if ring is None:
                        ring = ctx.ingest_ring(rgb.shape[0], rgb.shape[1], depth=16)
                    rgb = ring.push(rgb)
                    owned = True
                q.put((fi, rgb, owned))
            if ring is not None:
                ring.close()
        except BaseException as e:          # noqa: BLE001 -- re-raised below
            state["error"] = e
        finally:
            q.put(None)

    th = threading.Thread(target=reader, name="pvface-extract-reader")
    th.start()
    try:
        with open(landmark_output, 'wb') as flandmark, open(embedding_output, 'wb') as fembedding:
            pend_f, pend_b, pend_k, pend_own = [], [], [], []
            wq = queue.Queue(maxsize=4)

            def writer():
                # the text of batch i is formatted (in the library, outside the GIL) and written while the GPU computes batch i + 1
                try:
                    while True:
                        item = wq.get()
                        if item is None:
                            return
                        if state["error"] is not None:
                            continue
                        T, ident, pts, emb = item
                        flandmark.write(formats.landmark_rows(T, ident, pts, frame_width, frame_height))
                        fembedding.write(formats.embedding_rows(T, ident, emb))
                except BaseException as e:          # noqa: BLE001 -- re-raised below
                    state["error"] = e
                    while wq.get() is not None:
                        pass

            wth = threading.Thread(target=writer, name="pvface-extract-writer")
            wth.start()

            def flush():
                if not pend_b:
                    return
                pts, emb = ctx.landmarks_embed(pend_f, pend_b)
                wq.put(([k[0] for k in pend_k], [k[1] for k in pend_k], pts, emb))
                for f in pend_own:
                    f.release()
                del pend_f[:], pend_b[:], pend_k[:], pend_own[:]
            try:
                while state["error"] is None:
                    item = q.get()
                    if item is None:
                        break
                    fi, dev, owned = item
                    T, g = want[fi]
                    if owned:
                        pend_own.append(dev)
                    for ident, box in g:
                        pend_f.append(dev); pend_b.append(box); pend_k.append((T, ident))
                    if len(pend_b) >= batch:
                        flush()
                if state["error"] is None:
                    flush()
            finally:
                wq.put(None)
                wth.join()
    finally:
        while th.is_alive():                 # an error on this side: let the reader run out
            try:
                q.get(timeout=0.05)
            except queue.Empty:
                pass
        th.join()
    if state["error"] is not None:
        raise state["error"]


def process(video, shot, landmark_model, embedding_model, tracking_output, landmark_output, embedding_output, label_output=None,
            detect_min_size=0.0, detect_every=0.0, track_min_overlap_ratio=MIN_OVERLAP_RATIO, track_min_confidence=MIN_CONFIDENCE,
            track_max_gap=MAX_GAP, threshold=0.6, ctx=None):
    """`track` + `extract` (+ `cluster`) in ONE pass over the video -- one decode, one upload per frame; the reference needs two decodes
    (pyannote-face.py:261 and :287).  Writes the same three files as the separate verbs, line for line (the faces of one frame in the
    order pandas' sort of the complete track table gives them: formats.file_order), plus the `identifier label` file of `cluster`."""
    pipe = _pipeline(video, ctx, landmark_model, embedding_model, detect_min_size=detect_min_size, detect_every=detect_every,
                     track_min_overlap_ratio=track_min_overlap_ratio, track_min_confidence=track_min_confidence, track_max_gap=track_max_gap,
                     threshold=threshold)
    shots = load_shots(shot) if isinstance(shot, str) else shot
    state = {"next": 0}
    with open(tracking_output, 'w') as foutput:
        def write(tracks):
            for trk in tracks:
                for line in formats.track_lines(state["next"], trk):
                    foutput.write(line)
                foutput.flush()
                state["next"] += 1
        tm = {}
        res = pipe.run_stream(video, shots, on_tracks=write, cluster=label_output is not None, timings=tm)
        res["timings"] = tm
    w, h = video.size
    with open(landmark_output, 'wb') as flandmark, open(embedding_output, 'wb') as fembedding:
        flandmark.write(formats.landmark_rows(res["face_T"], res["face_id"], res["landmarks"], w, h))
        fembedding.write(formats.embedding_rows(res["face_T"], res["face_id"], res["embeddings"]))
    if label_output is not None:
        with open(label_output, 'w') as f:
            for identifier in sorted(set(res["face_id"].tolist())):
                f.write('{identifier:d} {label:d}\n'.format(identifier=identifier, label=res["labels"].get(identifier, identifier)))
    return res


def _times(video):
    n = len(video)
    return [(i / video.frame_rate, None) for i in range(n)]


def cluster(embeddings, output, threshold=0.6, force=False, metric="euclidean", ctx=None):
    """FaceClustering on an embedding file (face/clustering.py:130-134) -> `identifier label` lines for `demo --label`.
    Tracks that take no part in the clustering (a single timestamp: clustering.py:78-79) keep their own identifier as label."""
    from .clustering import FaceClustering
    clustering = FaceClustering(threshold=threshold, force=force, metric=metric, ctx=ctx)
    starting_point, features = clustering.model.preprocess(embeddings)
    result = clustering(starting_point, features=features)
    label = {int(track): int(lab) for _, track, lab in result.itertracks(yield_label=True)}
    with open(output, 'w') as f:
        for identifier in sorted(set(int(t) for t in np.unique(features.track))):
            f.write('{identifier:d} {label:d}\n'.format(identifier=identifier, label=label.get(identifier, identifier)))
    return label


def shot(video, output, height=50, window=2.0, threshold=1.0, ctx=None):
    """Shot boundary detection (scripts/pyannote-structure.py:65-70): a pyannote.core.json Timeline file that `track` reads back"""
    from .structure import Shot
    segments = sorted(Shot(video, height=height, context=window, threshold=threshold, ctx=ctx))
    with open(output, 'w') as fp:
        json.dump({"pyannote": "Timeline", "content": [{"start": s.start, "end": s.end} for s in segments]}, fp)
    return segments


def main(argv=None):
    ap = argparse.ArgumentParser(prog="pyannote-face", description="face tracking => feature extraction => face clustering (MI355X)")
    ap.add_argument("--fps", type=float, default=25.0, help="frame rate of a .npy / synthetic video")
    ap.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK or 0)")
    ap.add_argument("--metrics", default=None, help="write a JSON file with the run's timings and counts (frames, tracks, faces, seconds per stage, "
                    "frames resident at peak, device memory in use) next to the outputs")
    sub = ap.add_subparsers(dest="verb", required=True)
    t = sub.add_parser("track")
    t.add_argument("video"); t.add_argument("shot"); t.add_argument("tracking")
    t.add_argument("--min-size", type=float, default=0.0)
    t.add_argument("--every", type=float, default=0.0)
    t.add_argument("--min-overlap", type=float, default=MIN_OVERLAP_RATIO)
    t.add_argument("--min-confidence", type=float, default=MIN_CONFIDENCE)
    t.add_argument("--max-gap", type=float, default=MAX_GAP)
    e = sub.add_parser("extract")
    for name in ("video", "tracking", "landmark_model", "embedding_model", "landmarks", "embeddings"):
        e.add_argument(name)
    pr = sub.add_parser("process", help="track + extract + cluster in one pass over the video")
    for name in ("video", "shot", "landmark_model", "embedding_model", "tracking", "landmarks", "embeddings"):
        pr.add_argument(name)
    pr.add_argument("--labels", default=None)
    pr.add_argument("--min-size", type=float, default=0.0)
    pr.add_argument("--every", type=float, default=0.0)
    pr.add_argument("--min-overlap", type=float, default=MIN_OVERLAP_RATIO)
    pr.add_argument("--min-confidence", type=float, default=MIN_CONFIDENCE)
    pr.add_argument("--max-gap", type=float, default=MAX_GAP)
    pr.add_argument("--threshold", type=float, default=0.6)
    s = sub.add_parser("shot")
    s.add_argument("video"); s.add_argument("output")
    s.add_argument("--height", type=int, default=50, help="height of the images the optical flow runs on (reference default 50: one pyramid level; "
                   "64 and more bring OpenCV's coarser levels)")
    s.add_argument("--window", type=float, default=2.0)
    s.add_argument("--threshold", type=float, default=1.0)
    c = sub.add_parser("cluster")
    c.add_argument("embeddings"); c.add_argument("labels")
    c.add_argument("--threshold", type=float, default=0.6)
    c.add_argument("--force", action="store_true")
    c.add_argument("--metric", choices=("euclidean", "cosine"), default="euclidean")
    a = ap.parse_args(argv)
    import time
    ctx = None
    if a.device is not None:
        from .runtime import Context
        ctx = Context(device=a.device)
    t_begin = time.perf_counter()
    res = None
    if a.verb == "track":
        res = track(open_video(a.video, a.fps), a.shot, a.tracking, detect_min_size=a.min_size, detect_every=a.every,
                    track_min_overlap_ratio=a.min_overlap, track_min_confidence=a.min_confidence, track_max_gap=a.max_gap, ctx=ctx)
    elif a.verb == "process":
        res = process(open_video(a.video, a.fps), a.shot, a.landmark_model, a.embedding_model, a.tracking, a.landmarks, a.embeddings, a.labels,
                detect_min_size=a.min_size, detect_every=a.every, track_min_overlap_ratio=a.min_overlap,
                track_min_confidence=a.min_confidence, track_max_gap=a.max_gap, threshold=a.threshold, ctx=ctx)
    elif a.verb == "shot":
        shot(open_video(a.video, a.fps), a.output, height=a.height, window=a.window, threshold=a.threshold, ctx=ctx)
    elif a.verb == "extract":
        extract(open_video(a.video, a.fps), a.landmark_model, a.embedding_model, a.tracking, a.landmarks, a.embeddings, ctx=ctx)
    else:
        cluster(a.embeddings, a.labels, threshold=a.threshold, force=a.force, metric=a.metric, ctx=ctx)
    if a.metrics:
        from . import runtime
        m = {"verb": a.verb, "seconds": round(time.perf_counter() - t_begin, 4)}
        if isinstance(res, dict):
            m.update(frames=res.get("frames"), tracks=len(res.get("tracks", ())), faces=int(len(res["face_T"])) if "face_T" in res else None,
                     clusters=len(set(res["labels"].values())) if res.get("labels") else None, peak_frames_resident=res.get("peak_frames_resident"),
                     stage_seconds={k: round(v, 4) for k, v in res.get("timings", {}).items()})
        try:
            free, total = (ctx or runtime.default_context()).mem_info()
            m["device_memory_in_use_bytes"] = total - free
        except Exception:          # noqa: BLE001 -- the figure is optional
            pass
        with open(a.metrics, "w") as f:
            json.dump(m, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
