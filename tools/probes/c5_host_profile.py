#!/usr/bin/env python
"""dev-time probe: cProfile of the caller's thread (tracking state machine) for configs[4] (4K, 50 fps, 40 faces) and its GPU thread's idle share."""
import os, sys, time, tempfile, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import torch
from pyannote_video_amd import synth, models, pipeline
from pyannote_video_amd.runtime import Context

W, H, N, FACES, FPS, SHOTS, B = (int(os.environ.get("PW", 3840)), int(os.environ.get("PH", 2160)), int(os.environ.get("PN", 500)),
                                int(os.environ.get("PF", 40)), float(os.environ.get("PFPS", 50)), int(os.environ.get("PS", 2)), int(os.environ.get("PB", 32)))
dev = torch.device("cuda", 0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_rank0"), small=False)
video = synth.SyntheticVideo(width=W, height=H, n_frames=N, n_shots=SHOTS, faces=FACES, seed=20260925, frame_rate=FPS)
ft = video.frames_torch(dev); torch.cuda.synchronize()
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(N)]
times = [video.timestamp(i) for i in range(N)]
shots = video.shots()
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=B)
for it in range(2):
    tm = {}; ctx.sync(); t0 = time.perf_counter(); pipe.run(frames, times, video.frame_rate, shots, timings=tm, cluster=True)
    print("step %.1f ms" % ((time.perf_counter() - t0) * 1e3), {k: round(v * 1e3, 1) for k, v in tm.items() if isinstance(v, float)})
pr = cProfile.Profile(); tm = {}
ctx.sync(); pr.enable(); pipe.run(frames, times, video.frame_rate, shots, timings=tm, cluster=True); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print("\n".join(l[:170] for l in s.getvalue().split("\n")[4:46]))
