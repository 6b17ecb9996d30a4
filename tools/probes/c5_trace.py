#!/usr/bin/env python
"""dev-time probe: the engine's event trace for configs[4] (4K, 50 fps, 40 faces, windowed tracker starts)."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline
from pyannote_video_amd.runtime import Context
dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=3840, height=2160, n_frames=500, n_shots=2, faces=40, seed=20260925, frame_rate=50.0)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(500)]
times = [video.timestamp(i) for i in range(500)]
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=32)
for it in range(2):
    t0 = time.perf_counter(); pipe.run(frames, times, video.frame_rate, video.shots(), cluster=True); print("step %.1f ms" % ((time.perf_counter() - t0) * 1e3))
path = os.path.join(tempfile.gettempdir(), "pvf_trace.json")
os.environ["PVF_TRACE"] = path
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter(); pipe.run(frames, times, video.frame_rate, video.shots(), cluster=True); wall = time.perf_counter() - t0
ctx.prof_enable(False)
tot = sum(ctx.prof_get(n)[0] for n in ("pyramid", "fhog", "score", "chip", "ert", "conv", "dsst", "pdist", "hac"))
print("wall %.1f ms, kernels %.1f ms" % (wall * 1e3, tot))
ev = json.load(open(path)); t0 = ev[0][0]; prev = t0
for e in ev:
    print("%8.2f ms  (+%6.2f)  %s" % ((e[0] - t0) * 1e3, (e[0] - prev) * 1e3, " ".join(str(x) for x in e[1:]))); prev = e[0]
