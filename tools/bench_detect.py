"""Detector stage timings (HIP events on the library's stream) for one batch size: python tools/bench_detect.py [batch] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np
from pyannote_video_amd import models, pipeline, runtime
from pyannote_video_amd.synth import SyntheticVideo

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = runtime.Context(0, detector=models.DEFAULT_DETECTOR)
video = SyntheticVideo(n_frames=batch, height=1080, width=1920, n_shots=1, faces=8, seed=3)
frames = [ctx.stage(video.frame(i)) for i in range(batch)]
out = ctx.detect_batch(frames, 1)
nd = sum(len(o) for o in out)
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.time()
for _ in range(reps):
    out2 = ctx.detect_batch(frames, 1)
ctx.sync()
dt = time.time() - t0
assert sum(len(o) for o in out2) == nd
geo = pipeline.detector_geometry(1080, 1920)
flop = sum(g[4] for g in geo) * 3100 * 5 * 2.0 * batch * reps
line = ["batch %d reps %d dets %d wall %.1f ms/frame %.3f" % (batch, reps, nd, dt * 1e3, dt * 1e3 / (batch * reps))]
for fam in ("pyramid", "fhog_grad", "fhog_hist", "fhog_feat", "fhog", "score"):
    ms, n = ctx.prof_get(fam)
    line.append("%s %.3f ms/frame" % (fam, ms / (batch * reps)))
    if fam == "score" and ms > 0:
        line.append("score %.1f TFLOP/s" % (flop / (ms * 1e-3) / 1e12))
print(" | ".join(line))
