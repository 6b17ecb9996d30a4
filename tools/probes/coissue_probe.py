"""can the MFMA-bound embedder run INSIDE the VALU-bound pyramid kernels (designed pairing, not two streams and hope)?
Two contexts on one device, two Python threads:
  A  the image pyramids of a 125-frame 1080p batch (resize_rows_k x 20, the detector stream), `reps` times
  B  the embedding of 4096 chips (stem + conv3x3_c32_k + conv_mfma_k ..., the main stream), `reps` times
alone and together, with the stream priorities the engine uses (embedder above the detector), equal, and reversed (pyramid above the
embedder: six resize blocks fill a CU first and ONE embedder block fits beside them when the resize kernel is the thin variant --
PVF_RESIZE_THIN=1: 13.3 KB of LDS per block instead of 23.5).
    python tools/probes/coissue_probe.py [reps] [out.json]
PVF_RESIZE_THIN is read by the library at the first launch: run the script once per setting."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np  # noqa: E402
from pyannote_video_amd import models, runtime  # noqa: E402
from pyannote_video_amd.synth import SyntheticVideo  # noqa: E402
import tempfile  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out_path = sys.argv[2] if len(sys.argv) > 2 else None
PYR_PER_EMBED = 4          # four pyramid batches (4 x 6.2 ms) per embedding call (26 ms): the two sides take about the same time
only = os.environ.get("COISSUE_ONLY")      # e.g. "equal:together" -- one setting, one mode (for a counter pass)
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_bench"), small=True)
video = SyntheticVideo(n_frames=16, height=1080, width=1920, n_shots=1, faces=8, seed=3)
frames_np = [video.frame(i) for i in range(16)]
rng = np.random.default_rng(1)
chips = rng.integers(0, 256, (4096, 150, 150, 3), dtype=np.uint8)
res = {"reps": reps, "frames": 125, "chips": 4096, "pyramid_batches_per_embedding_call": 4, "resize_thin": os.environ.get("PVF_RESIZE_THIN", "0")}


def timed(fns):
    ths = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return time.perf_counter() - t0


for name, pa, pb in (("embedder_above_pyramid (the engine's setting)", -1, 1), ("equal", -1, -1), ("pyramid_above_embedder", 2, 2)):
    A = runtime.Context(0, detector=models.DEFAULT_DETECTOR, priority=pa)
    B = runtime.Context(0, detector=models.DEFAULT_DETECTOR, priority=pb)
    B.load_embedder(ep)
    fa = [A.upload(frames_np[i % 16]) for i in range(125)]

    def work_a(k=reps):
        for _ in range(k * PYR_PER_EMBED):
            A.pyramid_batch(fa, 1)

    def work_b(k=reps):
        for _ in range(k):
            B.embed_chips(chips)
        B.sync()
    work_a(1); work_b(1)
    if only:
        if only.split(":")[0] in name:
            mode = only.split(":")[1]
            timed({"together": [work_a, work_b], "pyramid": [work_a], "embed": [work_b]}[mode])
        A.close(); B.close()
        continue
    a = timed([work_a]) / reps * 1e3
    b = timed([work_b]) / reps * 1e3
    both = timed([work_a, work_b]) / reps * 1e3
    res[name] = {"pyramid_alone_ms": round(a, 2), "embed_alone_ms": round(b, 2), "together_ms": round(both, 2), "serial_ms": round(a + b, 2),
                 "together_over_serial": round(both / (a + b), 3)}
    A.close(); B.close()
print(json.dumps(res, indent=1))
if out_path:
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
