"""How much of the tracker / extraction work can hide beside the detector on ONE GPU?  (measured before building it.)

Two contexts on the same device (two HIP streams, the second with high priority), two Python threads (ctypes releases the GIL in every
library call):
  A  the detector on a 128-frame 1080p batch (pyramid + FHOG: VALU-bound; scoring: fp32 MFMA-bound), `reps` times
  B  the latency-bound half of the step: 2000 tracker starts + deferred updates (one CU-resident block per tracker) and the embedding
     of 2000 chips (fp32 MFMA), `reps` times
timed alone and together.  serial = A + B is what the single-stream engine pays today; `together` is what two streams would pay if the
host side were free.  gain = 1 - together / serial.
    python tools/probes/overlap_probe.py [reps] [out.json]
"""
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
import numpy as np  # noqa: E402
from pyannote_video_amd import models, runtime  # noqa: E402
from pyannote_video_amd.synth import SyntheticVideo  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out_path = sys.argv[2] if len(sys.argv) > 2 else None
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_bench"), small=True)
A = runtime.Context(0, detector=models.DEFAULT_DETECTOR)
B = runtime.Context(0, detector=models.DEFAULT_DETECTOR, priority=1)
B.load_embedder(ep)
batch = 128
video = SyntheticVideo(n_frames=batch, height=1080, width=1920, n_shots=1, faces=8, seed=3)
frames_np = [video.frame(i) for i in range(batch)]
fa = [A.stage(f) for f in frames_np]
fb0, fb1 = B.stage(frames_np[0]), B.stage(frames_np[1])
n = 2000
rng = np.random.default_rng(1)
boxes = []
for _ in range(n):
    s = float(rng.integers(80, 240)); x = float(rng.integers(0, 1920 - 240)); y = float(rng.integers(0, 1080 - 240))
    boxes.append((x, y, x + s, y + s))
trk = B.tracker_create_many(n)
chips = rng.integers(0, 256, (n, 150, 150, 3), dtype=np.uint8)


def work_a(k):
    for _ in range(k):
        A.detect_batch(fa, 1)
    A.sync()


def work_b(k, what=("trk", "emb")):
    for _ in range(k):
        if "trk" in what:
            B.tracker_start_many(trk, [fb0] * n, boxes)
            B.tracker_update_many(trk, [fb1] * n, defer=True)
        if "emb" in what:
            B.embed_chips(chips)
    B.sync()


def timed(fns):
    ths = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return time.perf_counter() - t0


work_a(1); work_b(1)                           # warm-up: plans, scratch, module load
res = {"reps": reps, "batch": batch, "trackers": n, "chips": n}
res["detector_alone_ms"] = round(timed([lambda: work_a(reps)]) / reps * 1e3, 2)
for name, what in (("tracker", ("trk",)), ("embed", ("emb",)), ("tracker_embed", ("trk", "emb"))):
    alone = timed([lambda: work_b(reps, what)]) / reps * 1e3
    both = timed([lambda: work_a(reps), lambda: work_b(reps, what)]) / reps * 1e3
    serial = res["detector_alone_ms"] + alone
    res[name] = {"alone_ms": round(alone, 2), "together_with_detector_ms": round(both, 2), "serial_ms": round(serial, 2),
                 "gain": round(1.0 - both / serial, 3)}
    print(name, res[name]); sys.stdout.flush()
# how often B fits beside ONE detector batch: B repeated while A runs once (B's work is the smaller half of a step)
print(json.dumps(res))
if out_path:
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
A.close(); B.close()
