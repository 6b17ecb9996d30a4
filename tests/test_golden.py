"""Golden fixtures (tests/golden/hotpath_small.npz, made by tests/golden/make_golden.py from the CPU oracle).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import os
import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_small.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(G)


def test_oracle_reproduces_golden(gold, oracle, model_paths):
    from pyannote_video_amd import models
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    f0 = gold["frames"][0]
    dets = det.detect(f0, 1)
    assert np.array_equal(np.array([d[5] for d in dets], np.int32), gold["boxes"])
    assert np.array_equal(np.array([d[0] for d in dets], np.float32), gold["scores"])
    assert np.array_equal(oracle.fhog(f0[20:148, 40:200], 8, 10, 10), gold["fhog_crop"])
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    assert np.array_equal(np.stack([sp(f0, b) for b in gold["boxes"]]), gold["landmarks"])
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    assert np.array_equal(emb.chip(f0, gold["landmarks"][0]), gold["chips"][0])
    assert np.abs(emb.forward(gold["chips"][0]) - gold["embeddings"][0]).max() < 1e-6
    tk = oracle.Tracker(models.dsst_tables())
    tk.start_track(f0, tuple(float(x) for x in gold["boxes"][0]))
    assert tk.update(gold["frames"][1]) == gold["tracker_psr"][0]
    assert tk.get_position() == tuple(gold["tracker_pos"][0])
    labels, _ = oracle.hac(oracle.pair_mean_dist(gold["clu_X"], gold["clu_row_start"]), gold["clu_sizes"], 0.6)
    assert np.array_equal(labels, gold["clu_labels"])


@pytest.mark.gpu
def test_hip_reproduces_golden(gold, ctx):
    f0 = gold["frames"][0]
    boxes, scores = ctx.detect(f0, 1)
    assert np.array_equal(np.array(boxes, np.int32), gold["boxes"])
    assert np.array_equal(scores, gold["scores"])
    assert np.array_equal(ctx.fhog(f0[20:148, 40:200], 8, 10, 10), gold["fhog_crop"])
    pts = ctx.landmarks([f0] * len(boxes), boxes)
    assert np.array_equal(pts, gold["landmarks"])
    assert np.array_equal(ctx.face_chips([f0] * len(boxes), pts), gold["chips"])
    e = ctx.embed([f0] * len(boxes), pts)
    assert np.linalg.norm(e - gold["embeddings"], axis=1).max() <= 1e-4
    t = ctx.tracker_create()
    ctx.tracker_start_many([t], [f0], [tuple(float(x) for x in gold["boxes"][0])])
    for i in (1, 2):
        psr, pos = ctx.tracker_update_many([t], [gold["frames"][i]])
        assert psr[0] == gold["tracker_psr"][i - 1]
        assert tuple(pos[0]) == tuple(gold["tracker_pos"][i - 1])
    ctx.tracker_destroy(t)
    labels, _ = ctx.cluster_tracks(gold["clu_X"], gold["clu_row_start"], 0.6)
    assert np.array_equal(labels, gold["clu_labels"])
    assert np.allclose(ctx.pair_mean_dist(gold["clu_X"], gold["clu_row_start"]), gold["clu_D"], rtol=1e-12, atol=1e-13)
