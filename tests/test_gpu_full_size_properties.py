"""GPU, at BASELINE.json configs[1]'s full size (1920x1080, 1000 frames, 4 shots, 8 faces per frame, full landmark model): properties
that do not need the CPU oracle (it would take ten minutes on this clip).
  * sharding is invisible: the video cut at a shot boundary into two frame ranges, each run on its own and stitched the way the multi-GPU
    path stitches (track-id offsets, the whole table's file order, one global clustering), gives the rows, ids, embeddings (bit for bit)
    and labels of the single run -- the decomposition bench.py --gpus N measures;
  * the run is deterministic: a second pass over the same frames returns the same bytes;
  * sanity against the generator's ground truth: one track per (shot, face), each starting on a generated face.
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
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "pyannote-video_amd"))
from pyannote_video_amd import synth, models, pipeline, formats, dist as pd
from pyannote_video_amd.runtime import Context
video = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925)
ft = video.frames_torch(torch.device("cuda", 0))
lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_fullsize"), small=False)
ctx = Context(device=0)
frames = [ctx.wrap_torch(ft[i]) for i in range(video.n_frames)]
times = [video.timestamp(i) for i in range(video.n_frames)]
shots = video.shots()
pipe = pipeline.FacePipeline(ctx, lp, ep, detect_batch_size=128)
whole = pipe.run(frames, times, video.frame_rate, shots)
again = pipe.run(frames, times, video.frame_rate, shots)
ranges = pipeline.split_into_shots(times, shots)
parts = []
for s0, s1 in ((0, 2), (2, 4)):
    i0, i1 = ranges[s0][0], ranges[s1 - 1][1]
    parts.append(pipe.run(frames[i0:i1], times[i0:i1], video.frame_rate, shots[s0:s1], cluster=False, last_shard=(s1 == 4), reorder=False))
off = [0, len(parts[0]["tracks"])]
T = np.concatenate([p["face_T"] for p in parts]); ids = np.concatenate([p["face_id"] + o for p, o in zip(parts, off)])
X = np.concatenate([p["X"] for p in parts]); E = np.concatenate([p["embeddings"] for p in parts]); L = np.concatenate([p["landmarks"] for p in parts])
fT = np.concatenate([p["file_T"] for p in parts]); fid = np.concatenate([p["file_id"] + o for p, o in zip(parts, off)])
perm = formats.file_order(T, ids, fT, fid)
T, ids, X, E, L = T[perm], ids[perm], X[perm], E[perm], L[perm]
labels = pd.global_cluster(pipe.clustering, T, ids, E)
out = {
    "tracks_whole": len(whole["tracks"]), "tracks_parts": [len(p["tracks"]) for p in parts], "faces": int(len(whole["face_T"])),
    "rows_equal": bool(np.array_equal(T, whole["face_T"]) and np.array_equal(ids, whole["face_id"])),
    "embeddings_equal": bool(np.array_equal(E, whole["embeddings"]) and np.array_equal(X, whole["X"])),
    "landmarks_equal": bool(np.array_equal(L, whole["landmarks"])),
    "labels_equal": labels == whole["labels"],
    "deterministic": bool(np.array_equal(again["embeddings"], whole["embeddings"]) and np.array_equal(again["landmarks"], whole["landmarks"])
                          and again["labels"] == whole["labels"] and again["tracks"] == whole["tracks"]),
    "clusters": len(set(whole["labels"].values())), "shots_x_faces": sum(len(s) for s in video.tracks),
    "identities": len(set(tr["ident"] for s in video.tracks for tr in s)),
}
# ground truth: every track starts on one of the generator's faces (centre of its first box within a few pixels of a face centre)
hits = 0
for tr in whole["tracks"]:
    t0, box = tr[0][0], tr[0][1]
    f = int(round(t0 * video.frame_rate)); k = video.shot_of(f); j = f - video.shot_bounds[k]
    cx, cy = (box[0] + box[2]) / 2 * 1920, (box[1] + box[3]) / 2 * 1080
    d = min(((g["cx"][j] - cx) ** 2 + (g["cy"][j] - cy) ** 2) ** 0.5 for g in video.tracks[k])
    hits += d < 12.0
out["tracks_on_faces"] = int(hits)
print("RESULT " + json.dumps(out))
ctx.close()
'''


def test_full_size_sharding_invisible_and_deterministic(tmp_path):
    script = tmp_path / "fullsize.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["tracks_whole"] == sum(r["tracks_parts"]) == r["shots_x_faces"] == 32, r
    assert r["faces"] > 7900
    assert r["rows_equal"] and r["embeddings_equal"] and r["landmarks_equal"] and r["labels_equal"], r
    assert r["deterministic"], r
    assert r["tracks_on_faces"] == 32 and 1 <= r["clusters"] <= 32, r
