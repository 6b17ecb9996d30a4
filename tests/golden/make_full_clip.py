#!/usr/bin/env python
"""Whole-clip fixtures: the CPU oracle flow (oracle/ref_flow.py) over EVERY frame of a clip bench.py times, frozen as
tests/golden/<name>.npz (layout: oracle/golden.py).  Needs no GPU and no /root/reference; minutes of CPU per clip:

    python tests/golden/make_full_clip.py c2_full      # BASELINE.json configs[1]: 1000 frames 1080p (about 20-40 min on 8 cores)
    python tests/golden/make_full_clip.py c4_clip0     # configs[3]: clip 0 of the 720p farm, 250 frames
    python tests/golden/make_full_clip.py c3_clip0     # configs[2]: the first 1000-frame clip of the long streamed video
    python tests/golden/make_full_clip.py c5_shot0     # configs[4]: the first shot (250 frames) of the 4K crowd clip (about an hour)

The frames are `SyntheticVideo.frame(i)`, byte-identical to what `frames_torch` puts into HBM for the bench (tests/test_engine.py pins
that); the models are the seeded synthetic ones of `models.ensure_synthetic_models` (full 15 x 500 x 500 landmark model).  What follows
the reference: scripts/pyannote-face.py:239-314 (track, extract), pyannote/video/tracking.py:331-357,374-434, face/clustering.py:92-119.
PARITY UNPINNED like the oracle (dlib / pyannote.algorithms absent): this freezes the restated algorithms.
"""
import concurrent.futures
import ctypes as C
import os
import sys
import tempfile
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
from pyannote_video_amd import synth, models, pipeline  # noqa: E402
from oracle import oracle, ref_flow, golden  # noqa: E402


class RecordingDetector(object):
    """the oracle detector, one scan per frame: keeps the raw candidates (before NMS) and returns what pvo_detect would"""

    def __init__(self, det):
        self.det, self.raw = det, []

    def __call__(self, rgb):
        raw = self.det.detect_raw(rgb, 1)
        sc = np.array([d[0] for d in raw], np.float32)
        self.raw.append(golden.raw_key([d[2] for d in raw], [d[1] for d in raw], [d[3] for d in raw], [d[4] for d in raw], sc.view(np.int32)))
        n = len(raw)
        buf = (oracle._Det * max(n, 1))()
        for i, d in enumerate(raw):
            buf[i] = oracle._Det(d[0], d[1], d[2], d[3], d[4], *d[5])
        out = (oracle._Det * max(n, 1))()
        k = oracle.lib().pvo_nms(buf, n, C.c_double(self.det.s.nms_iou), C.c_double(self.det.s.nms_covered), out, max(n, 1))
        return [(out[i].l, out[i].t, out[i].rr, out[i].b) for i in range(k)]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2_full"
    va = golden.CLIPS[name]
    v, take = golden.video_of(name)
    threads = oracle.usable_cpus(cap=1024)
    oracle.lib().pvo_set_threads(threads)
    lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_golden"), small=False)
    det = RecordingDetector(oracle.Detector(models.load_container(models.DEFAULT_DETECTOR)))
    sp = oracle.ShapePredictor(models.load_model_file(lp, "shape_predictor"))
    emb = oracle.Embedder(models.load_model_file(ep, "embedder"))
    tabs = models.dsst_tables()
    # one scan must equal pvo_detect (raw + NMS in one call)
    f0 = v.frame(0)
    assert det(f0) == oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))(f0)
    det.raw = []
    t0 = time.perf_counter()
    frames = [v.frame(i) for i in range(take)]
    times = [v.timestamp(i) for i in range(take)]
    shots = golden.shots_of(v, take)
    print("%s: %d frames rendered (%.0f s), oracle flow on %d threads" % (name, len(frames), time.perf_counter() - t0, threads), flush=True)
    pool = concurrent.futures.ThreadPoolExecutor(min(threads, 32)) if threads > 1 else None
    t0 = time.perf_counter()
    tracks = ref_flow.track_video(frames, times, shots, det, lambda: oracle.Tracker(tabs), v.frame_rate,
                                  min_conf=pipeline.CLI_MIN_CONFIDENCE, ratio=pipeline.CLI_MIN_OVERLAP_RATIO, max_gap=pipeline.CLI_MAX_GAP, pool=pool)
    print("  detect + tracking: %.0f s, %d tracks" % (time.perf_counter() - t0, len(tracks)), flush=True)
    keep = []
    lm, em = ref_flow.extract(ref_flow.track_text(tracks), frames, times, sp, emb, pool=pool, keep=keep)
    labels = ref_flow.cluster(em, 0.6)
    dt = time.perf_counter() - t0
    print("  whole flow: %.0f s, %d faces, %d clusters" % (dt, len(em), len(set(labels.values()))), flush=True)
    assert len(det.raw) == take
    g = golden.pack(va, tracks, lm, keep, labels, det.raw, v.frame_rate, v.frame_size, seconds=dt, threads=threads)
    np.savez_compressed(golden.path(name), **g)
    print("wrote", golden.path(name), os.path.getsize(golden.path(name)), "bytes")
    # the fixture reproduces what it was made from
    z = golden.load(name)
    assert golden.tracks_of(z) == tracks


if __name__ == "__main__":
    main()
