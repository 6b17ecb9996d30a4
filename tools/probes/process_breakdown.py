#!/usr/bin/env python
"""dev-time probe: where the one-pass `process` verb spends its time beyond the engine run (configs[1] clip, frames resident)."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
import bench
from pyannote_video_amd import synth, models, cli, formats
from pyannote_video_amd.runtime import Context
from pyannote_video_amd._core import Segment

dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(1000)]
rv = bench.ResidentVideo(frames, video.frame_rate, video.frame_size)
shots = [Segment(a, b) for a, b in video.shots()]
d = tempfile.mkdtemp()
P = lambda k: os.path.join(d, k + ".txt")
orig_rs = cli._pipeline
for it in range(3):
    marks = {}
    import pyannote_video_amd.pipeline as pl
    orig = pl.FacePipeline.run_stream
    def timed(self, *a, **k):
        t0 = time.perf_counter(); r = orig(self, *a, **k); marks["run_stream"] = time.perf_counter() - t0; return r
    pl.FacePipeline.run_stream = timed
    olr, oer = formats.landmark_rows, formats.embedding_rows
    def tl(*a, **k):
        t0 = time.perf_counter(); r = olr(*a, **k); marks["landmark_rows"] = time.perf_counter() - t0; return r
    def te(*a, **k):
        t0 = time.perf_counter(); r = oer(*a, **k); marks["embedding_rows"] = time.perf_counter() - t0; return r
    formats.landmark_rows, formats.embedding_rows = tl, te
    otl = formats.track_lines
    acc = {"t": 0.0}
    def ttl(i, trk):
        t0 = time.perf_counter(); r = list(otl(i, trk)); acc["t"] += time.perf_counter() - t0; return r
    formats.track_lines = ttl
    ctx.sync(); t0 = time.perf_counter()
    res = cli.process(rv, shots, lp, ep, P("t"), P("l"), P("e"), P("lab"), ctx=ctx)
    total = time.perf_counter() - t0
    pl.FacePipeline.run_stream = orig; formats.landmark_rows, formats.embedding_rows, formats.track_lines = olr, oer, otl
    print("process %.1f ms: run_stream %.1f (track_lines inside %.1f) landmark_rows %.1f embedding_rows %.1f timings %s" % (
        total * 1e3, marks["run_stream"] * 1e3, acc["t"] * 1e3, marks["landmark_rows"] * 1e3, marks["embedding_rows"] * 1e3,
        {k: round(v, 4) for k, v in res["timings"].items() if isinstance(v, float)}))
