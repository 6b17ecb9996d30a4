"""Shot-detector micro-bench: displaced frame differences of n resident 1080p frames: python tools/bench_shot.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np
import torch
torch.cuda.set_device(0)
from pyannote_video_amd import runtime, structure, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = runtime.Context(0)
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=n, n_shots=4, faces=8, seed=5)
ft = video.frames_torch(torch.device("cuda", 0))
frames = [ctx.wrap_torch(ft[i]) for i in range(n)]
t = structure.shot_tables()
ctx.shot_dfd(frames, 50, 88, t)
t0 = time.time(); d = ctx.shot_dfd(frames, 50, 88, t); dt = time.time() - t0
print("shot_dfd: %d frames 1080p in %.2f ms (%.0f frames/s); dfd at the cuts %s, median elsewhere %.2f"
      % (n, dt * 1e3, n / dt, [round(float(d[b - 1]), 1) for b in video.shot_bounds[1:-1]], float(np.median(d))))
shots = list(structure.Shot(video, ctx=ctx))
print("Shot(video): boundaries at frames", [round(s.end * video.frame_rate) for s in shots[:-1]], "true cuts", list(video.shot_bounds[1:-1]))
