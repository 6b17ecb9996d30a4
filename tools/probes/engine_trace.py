#!/usr/bin/env python
"""dev-time probe: the engine's own event trace (PVF_TRACE) of one configs[1] step, GPU-thread and caller-thread events with their gaps."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline
from pyannote_video_amd.runtime import Context
dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(1000)]
times = [video.timestamp(i) for i in range(1000)]
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=128)
for it in range(3):
    pipe.run(frames, times, video.frame_rate, video.shots(), cluster=True)
path = os.path.join(tempfile.gettempdir(), "pvf_trace.json")
os.environ["PVF_TRACE"] = path
ctx.prof_reset(); ctx.prof_enable(True)
pipe.run(frames, times, video.frame_rate, video.shots(), cluster=True)
ctx.prof_enable(False)
ev = json.load(open(path))
t0 = ev[0][0]
prev = t0
for e in ev:
    print("%8.2f ms  (+%6.2f)  %s" % ((e[0] - t0) * 1e3, (e[0] - prev) * 1e3, " ".join(str(x) for x in e[1:])))
    prev = e[0]
