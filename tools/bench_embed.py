"""Embedding-net micro-bench: n random 150x150 chips through the ResNet: python tools/bench_embed.py [n] [reps]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np
from pyannote_video_amd import models, runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = runtime.Context(0)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_bench"), small=True)
ctx.load_embedder(ep)
chips = np.random.default_rng(0).integers(0, 256, (n, 150, 150, 3), dtype=np.uint8)
e0 = ctx.embed_chips(chips)
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.time()
for _ in range(reps):
    e = ctx.embed_chips(chips)
ctx.sync()
dt = (time.time() - t0) / reps
ms, k = ctx.prof_get("conv")
assert np.array_equal(e, e0)
import zlib
print("embed: n %d wall %.2f ms, conv family %.2f ms per call (%d launches) => %.1f TFLOP/s, crc32 of the descriptors %08x"
      % (n, dt * 1e3, ms / reps, k // reps, n * 0.5418e9 / (ms / reps * 1e-3) / 1e12, zlib.crc32(e.tobytes())))
