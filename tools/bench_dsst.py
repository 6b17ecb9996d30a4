"""Correlation-tracker micro-bench: n trackers started on one frame and updated (deferred) on the next: python tools/bench_dsst.py [n] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np
from pyannote_video_amd import models, runtime
from pyannote_video_amd.synth import SyntheticVideo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = runtime.Context(0, detector=models.DEFAULT_DETECTOR)
video = SyntheticVideo(n_frames=2, height=1080, width=1920, n_shots=1, faces=8, seed=3)
f0, f1 = ctx.stage(video.frame(0)), ctx.stage(video.frame(1))
rng = np.random.default_rng(1)
boxes = []
for _ in range(n):
    s = float(rng.integers(80, 240)); x = float(rng.integers(0, 1920 - 240)); y = float(rng.integers(0, 1080 - 240))
    boxes.append((x, y, x + s, y + s))
trk = ctx.tracker_create_many(n)
ctx.tracker_start_many(trk, [f0] * n, boxes)
ctx.tracker_update_many(trk, [f1] * n, defer=True)
ctx.sync()
tot = {"start": [0.0, 0.0, 0], "update(deferred)": [0.0, 0.0, 0]}
for _ in range(reps):
    for name, fn in (("start", lambda: ctx.tracker_start_many(trk, [f0] * n, boxes)),          # a restart drops the pending update
                     ("update(deferred)", lambda: ctx.tracker_update_many(trk, [f1] * n, defer=True))):
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.time()
        fn()
        ctx.sync()
        tot[name][0] += time.time() - t0
        ms, k = ctx.prof_get("dsst")
        ctx.prof_enable(False)
        tot[name][1] += ms; tot[name][2] = k
for name, (dt, ms, k) in tot.items():
    print("%s: n %d wall %.2f ms, dsst kernels %.2f ms per call (%d launches)" % (name, n, dt / reps * 1e3, ms / reps, k))
