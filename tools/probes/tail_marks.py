#!/usr/bin/env python
"""dev-time probe: host timestamps between the return of a step's last extraction call and the entry of its clustering call."""
import os, sys, time, tempfile, functools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline, engine, runtime, dist as pdist
from pyannote_video_amd.runtime import Context

marks = []
def mark(name):
    marks.append((time.perf_counter(), name))
def wrap(obj, attr, label=None):
    f = getattr(obj, attr)
    @functools.wraps(f)
    def g(*a, **k):
        mark((label or attr) + " >")
        try:
            return f(*a, **k)
        finally:
            mark((label or attr) + " <")
    setattr(obj, attr, g)

wrap(Context, "landmarks_embed")
wrap(engine.ExtractStream, "finish")
wrap(engine.ExtractStream, "plan_finish")
wrap(pipeline.FacePipeline, "_result")
wrap(engine.Engine, "run", "engine.run")
wrap(pdist, "gather_rows")
wrap(pdist, "global_cluster")
wrap(Context, "cluster_tracks_f32")

dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(1000)]
times = [video.timestamp(i) for i in range(1000)]
shots = video.shots()
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=128)
pipe.return_table = False

def step():
    res = pipe.run(frames, times, video.frame_rate, shots, cluster=False)
    T, ids, X, offsets = pdist.gather_rows(res["face_T"], res["face_id"], res["embeddings"], len(res["tracks"]), device=dev)
    return pdist.global_cluster(pipe.clustering, T, ids, X)

for it in range(4):
    del marks[:]
    ctx.sync(); t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    last = max(i for i, (t, n) in enumerate(marks) if n == "landmarks_embed <")
    base = marks[last][0]
    print("step %.1f ms; after the last extraction call returned (ms):" % ((t1 - t0) * 1e3))
    for t, n in marks[last:]:
        print("   +%7.3f  %s" % ((t - base) * 1e3, n))
    print("   +%7.3f  step returns" % ((t1 - base) * 1e3))
