"""The detector alone on one resident batch: HIP-event time of its kernel families per batch, nothing beside it on the GPU.
    python tools/bench_detector.py [frames=125] [reps=5] [height=1080] [width=1920]
PVF_DETECTOR_SCREENING=0 times the dense scoring kernel instead of the screening pass; PVF_LIBRARY selects another build of the library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
from pyannote_video_amd import models, runtime  # noqa: E402
from pyannote_video_amd.synth import SyntheticVideo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
ctx = runtime.Context(0, detector=models.DEFAULT_DETECTOR)
video = SyntheticVideo(n_frames=min(n, 16), height=h, width=w, n_shots=1, faces=8, seed=3)
frames = [ctx.upload(video.frame(i % min(n, 16))) for i in range(n)]
ctx.detect_many(frames, n, 1, arrays=True)
ctx.sync()
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(reps):
    ctx.detect_many(frames, n, 1, arrays=True)
ctx.sync()
ctx.prof_enable(False)
out = {"frames": n, "reps": reps, "frame": "%dx%d" % (w, h)}
for fam in ("pyramid", "fhog", "score", "score_screened"):
    ms, k = ctx.prof_get(fam)
    if k:
        out[fam + "_ms_per_batch"] = round(ms / k, 4)
out["screening"] = ctx.detector_screening_stats()
print(json.dumps(out))
