#!/usr/bin/env python
"""Generates tests/golden/hotpath_small.npz from the CPU oracle (the reference ships no vectors; dlib is not installable
here, so these freeze the restated algorithms -- PARITY UNPINNED, see oracle/pvo.h).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
from pyannote_video_amd import synth, models  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    v = synth.SyntheticVideo(width=400, height=240, n_frames=3, n_shots=1, faces=2, min_face=50, max_face=90, seed=3)
    frames = np.stack([v.frame(i) for i in range(3)])
    d = tempfile.mkdtemp()
    lp, ep = models.ensure_synthetic_models(d, small=True)
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_container(lp))
    emb = oracle.Embedder(models.load_container(ep))
    dets = det.detect(frames[0], 1)
    boxes = np.array([x[5] for x in dets], np.int32)
    scores = np.array([x[0] for x in dets], np.float32)
    pts = np.stack([sp(frames[0], b) for b in boxes])
    chips = np.stack([emb.chip(frames[0], p) for p in pts])
    embs = np.stack([emb.forward(c) for c in chips])
    fh = oracle.fhog(frames[0][20:148, 40:200], 8, 10, 10)
    tk = oracle.Tracker(models.dsst_tables())
    tk.start_track(frames[0], tuple(float(x) for x in boxes[0]))
    psr, pos = [], []
    for i in (1, 2):
        psr.append(tk.update(frames[i])); pos.append(tk.get_position())
    rng = np.random.default_rng(12)
    sizes = rng.integers(1, 6, 14)
    cent = rng.normal(size=(4, 128)); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    X = np.concatenate([np.round(0.55 * (cent[t % 4] + 0.04 * rng.normal(size=(sizes[t], 128))), 5) for t in range(14)])
    rs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    D = oracle.pair_mean_dist(X, rs)
    labels, _ = oracle.hac(D, sizes, 0.6)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_small.npz")
    np.savez_compressed(out, frames=frames, boxes=boxes, scores=scores, landmarks=pts, chips=chips, embeddings=embs,
                        fhog_crop=fh, tracker_psr=np.array(psr), tracker_pos=np.array(pos), clu_X=X, clu_row_start=rs,
                        clu_sizes=sizes, clu_D=D, clu_labels=labels)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
