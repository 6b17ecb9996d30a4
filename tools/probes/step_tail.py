#!/usr/bin/env python
"""dev-time probe: host time between the last embedding kernel and the end of a configs[1] step (cProfile of the caller's thread)."""
import os, sys, time, tempfile, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline, dist as pdist
from pyannote_video_amd.runtime import Context

dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(1000)]
times = [video.timestamp(i) for i in range(1000)]
shots = video.shots()
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=128)

def step(tm):
    res = pipe.run(frames, times, video.frame_rate, shots, timings=tm, cluster=False)
    t0 = time.perf_counter()
    T, ids, X, offsets = pdist.gather_rows(res["face_T"], res["face_id"], res["embeddings"], len(res["tracks"]), device=dev)
    t1 = time.perf_counter()
    labels = pdist.global_cluster(pipe.clustering, T, ids, X)
    tm["gather_s"] = t1 - t0; tm["cluster_s2"] = time.perf_counter() - t1
    return labels

for it in range(3):
    tm = {}; ctx.sync(); t0 = time.perf_counter(); step(tm); print("step %.1f ms" % ((time.perf_counter() - t0) * 1e3), {k: round(v * 1e3, 2) for k, v in tm.items()})
pr = cProfile.Profile(); tm = {}
ctx.sync(); pr.enable(); step(tm); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("cumulative"); ps.print_stats(70)
keep = ("finish", "_result", "round_rows", "gather_rows", "global_cluster", "preprocess", "cluster_arrays", "cluster_tracks", "file_order", "concatenate",
        "pandas_sort", "itertracks", "pair_mean", "cluster_dist", "run", "compute", "landmarks_embed", "asarray", "__call__", "ncalls")
for line in s.getvalue().split("\n"):
    if any(k in line for k in keep):
        print(line[:160])
