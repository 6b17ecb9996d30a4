"""Landmark micro-bench: n faces (full 15 x 500 x 500 model) on one resident 1080p frame: python tools/bench_ert.py [n] [reps]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np
from pyannote_video_amd import models, runtime
from pyannote_video_amd.synth import SyntheticVideo

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_bench_full"), small=False)
ctx = runtime.Context(0, landmarks=lp)
video = SyntheticVideo(n_frames=1, height=1080, width=1920, n_shots=1, faces=8, seed=3)
f = ctx.upload(video.frame(0))
rng = np.random.default_rng(1)
boxes = []
for _ in range(n):
    s = int(rng.integers(80, 240)); x = int(rng.integers(0, 1920 - 240)); y = int(rng.integers(0, 1080 - 240))
    boxes.append((x, y, x + s, y + s))
p0 = ctx.landmarks([f] * n, boxes)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(reps):
    p = ctx.landmarks([f] * n, boxes)
ctx.sync()
ms, k = ctx.prof_get("ert")
assert np.array_equal(p, p0)
print("ert: n %d, %.3f ms per call (%d launches), checksum %d" % (n, ms / reps, k // reps, int(p.astype(np.int64).sum())))
