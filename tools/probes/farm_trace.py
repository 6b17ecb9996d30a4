#!/usr/bin/env python
"""dev-time probe: the engine's event trace (PVF_TRACE) for a small clip farm (8 clips x 250 frames 720p), with the GPU-thread gaps."""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline
from pyannote_video_amd.runtime import Context
dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
ctx = Context(device=0)
clips, keep = [], []
for k in range(8):
    v = synth.SyntheticVideo(width=1280, height=720, n_frames=250, n_shots=2, faces=8, seed=100 + k, frame_rate=25.0)
    ft = v.frames_torch(dev); keep.append(ft)
    clips.append(dict(frames=[ctx.wrap_torch(ft[i]) for i in range(250)], times=[v.timestamp(i) for i in range(250)], frame_rate=25.0, shots=v.shots()))
torch.cuda.synchronize()
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=128)
for it in range(2):
    t0 = time.perf_counter(); pipe.run_many(clips); print("farm %.1f ms" % ((time.perf_counter() - t0) * 1e3))
path = os.path.join(tempfile.gettempdir(), "pvf_trace.json")
os.environ["PVF_TRACE"] = path
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter(); pipe.run_many(clips); wall = time.perf_counter() - t0
ctx.prof_enable(False)
tot = 0.0
for name in ("pyramid", "fhog", "score", "chip", "ert", "conv", "dsst", "pdist", "hac"):
    ms, n = ctx.prof_get(name); tot += ms
print("wall %.1f ms, kernels %.1f ms" % (wall * 1e3, tot))
ev = json.load(open(path))
t0 = ev[0][0]; prev = t0
for e in ev[:90]:
    print("%8.2f ms  (+%6.2f)  %s" % ((e[0] - t0) * 1e3, (e[0] - prev) * 1e3, " ".join(str(x) for x in e[1:])))
    prev = e[0]
