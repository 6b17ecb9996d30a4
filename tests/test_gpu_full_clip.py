"""GPU: the WHOLE benched clip against the CPU oracle, every frame of it.

tests/golden/c2_full.npz (BASELINE.json configs[1]: 1920x1080, 1000 frames, 4 shots, 8 faces, seed 20260925 -- the clip bench.py
times), c4_clip0.npz (configs[3]: clip 0 of the 720p farm), c3_clip0.npz (configs[2]: the first 1000-frame clip of the long video) and
c5_shot0.npz (configs[4]: the first 250-frame shot of the 4K / 40 faces clip) hold what the CPU oracle flow returns for ALL their frames
(tests/golden/make_full_clip.py, oracle/golden.py): every track row, every face row with its 68 points and its descriptor, every
cluster label, and per frame the detector's raw candidates before non-maximum suppression.  The product must reproduce all of it --
with the screening pass (the default: a data-dependent filter decides which windows get the exact arithmetic) and without:
  * tracks, face rows, landmarks, labels: exact;  descriptors: L2 <= 1e-4 (north_star's tolerance);
  * raw candidates (level, filter, row, column, score bits) of every frame: exact.
What this pins that the windowed checks could not: track ids are decided over the whole shot graph (reference tracking.py:331-357,
374-434) and cluster labels over all tracks (face/clustering.py:92-119).

Runs in a process of its own: the frames are synthesised on the device with torch, whose HIP runtime has to initialise before the library's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, os, sys, tempfile
import numpy as np
import torch
torch.cuda.set_device(0)
root, name = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "pyannote-video_amd"))
from pyannote_video_amd import synth, models, pipeline
from pyannote_video_amd.runtime import Context
from oracle import golden
g = golden.load(name)
video, take = golden.video_of(name)
ft = video.frames_torch(torch.device("cuda", 0), indices=range(take))
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_fullclip"), small=False)
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(take)]
times = [video.timestamp(i) for i in range(take)]
batch = 128 if video.size[0] <= 1920 else 32
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=batch)
out = {}
for leg, on in (("screened", True), ("dense", False)):
    ctx.detector_screening(on)
    res = pipe.run(frames, times, video.frame_rate, golden.shots_of(video, take))
    c = golden.compare(g, res)
    raw = ctx.detect_raw_many(frames, 125 if batch == 128 else batch)
    c["raw_candidates"] = golden.compare_raw(g, [golden.raw_key(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]) for r in raw])
    out[leg] = c
out["screening"] = ctx.detector_screening_stats()
out["raw_total"] = int(g["raw_counts"].sum())
print("RESULT " + json.dumps(out))
'''


def _run(name):
    sys.path.insert(0, ROOT)
    from oracle import golden
    if not golden.available(name):
        pytest.skip("tests/golden/%s.npz is not there (python tests/golden/make_full_clip.py %s)" % (name, name))
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, name], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("name", ["c2_full", "c4_clip0", "c3_clip0", "c5_shot0"])
def test_whole_clip_equals_the_oracle_flow(name):
    out = _run(name)
    for leg in ("screened", "dense"):
        c = out[leg]
        assert c["tracks"] == "exact", (leg, c["tracks"])
        assert c["face_rows"] == "exact", leg
        assert c["landmarks"] == "exact", (leg, c["landmarks"])
        assert c["labels"] == "exact", leg
        assert c["embed_l2_max"] is not None and c["embed_l2_max"] <= 1e-4, (leg, c["embed_l2_max"])      # north_star: embedding L2 within 1e-4
        assert c["raw_candidates"] == "exact", (leg, c["raw_candidates"])
        assert c["all_exact"]
    assert out["screening"]["retries"] == 0            # the screened leg really was screened (a retry would have run it dense)
    assert out["raw_total"] > 0
