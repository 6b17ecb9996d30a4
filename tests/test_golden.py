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


def test_whole_clip_fixtures_are_the_oracles_output_on_sampled_frames(oracle):
    """tests/golden/c2_full.npz / c4_clip0.npz (oracle/golden.py; the GPU tests and bench.py compare the product with them over ALL frames)
    re-derived from the oracle on a few frames here: the detector's raw candidates of the frame, and the 68 points + descriptor of the
    frame's first face row.  Pins the fixture files to the oracle sources of this tree (a changed oracle without a regenerated fixture
    fails here, on CPU)."""
    import os, sys, tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import golden
    from pyannote_video_amd import synth, models
    lp, ep = models.ensure_synthetic_models(os.path.join(tempfile.gettempdir(), "pvface_models_golden"), small=False)
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_model_file(lp, "shape_predictor"))
    emb = oracle.Embedder(models.load_model_file(ep, "embedder"))
    checked = 0
    for name, picks in (("c2_full", (0, 613)), ("c4_clip0", (124,)), ("c3_clip0", (377,)), ("c5_shot0", (101,))):
        if not golden.available(name):
            continue
        g = golden.load(name)
        v, take = golden.video_of(name)
        assert list(g["video"][:5]) == [v.size[0], v.size[1], take, v.n_shots, v.faces]
        assert len(g["raw_counts"]) == take and int(g["raw_counts"].sum()) == len(g["raw_rows"])
        tracks = golden.tracks_of(g)
        assert sum(len(t) for t in tracks) == len(g["track_rows"])
        w, h = v.frame_size
        for i in picks:
            f = v.frame(i)
            raw = det.detect_raw(f, 1)
            sc = np.array([d[0] for d in raw], np.float32)
            rows = golden.raw_key([d[2] for d in raw], [d[1] for d in raw], [d[3] for d in raw], [d[4] for d in raw], sc.view(np.int32))
            assert golden.raw_digest(rows) == g["raw_digest"][i], (name, i)
            k = int(np.nonzero(g["face_frame"] == i)[0][0])                  # first face row of the frame
            ident = int(g["face_id"][k])
            row = [r for r in g["track_rows"].tolist() if r[0] == i and r[1] == ident][0]
            # the box extract() hands to the landmark model: the track file's 3-decimal float32 value times the frame size, truncated
            box = tuple(int(float(np.float32("%.3f" % (c / s))) * s) for c, s in zip(row[2:6], (w, h, w, h)))
            pts = sp(f, box)
            assert np.array_equal(pts, g["landmarks"][k].astype(np.int32)), (name, i)
            assert np.array_equal(emb(f, pts).view(np.uint32), g["embeddings"][k].view(np.uint32)), (name, i)
            checked += 1
    if not checked:
        import pytest
        pytest.skip("no whole-clip fixture in tests/golden")


def test_fixture_comparison_of_a_longer_run():
    """oracle/golden.py compare(prefix=True): a fixture of a run's first shots against the longer run's result -- the fixture's tracks are
    the run's first ones, the faces before the fixture's last frame are compared in (frame, track) order (the order inside a timestamp
    is an artefact of pandas' unstable sort of the whole table), later tracks / faces and the labels are left alone; any difference inside
    the covered part is reported"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import golden
    if not golden.available("c4_clip0"):
        import pytest
        pytest.skip("no fixture")
    g = golden.load("c4_clip0")
    fr = float(g["frame_rate"])
    tracks = golden.tracks_of(g)
    n = len(g["face_frame"])
    rng = np.random.default_rng(3)
    # the longer run: the clip's faces with the rows of every timestamp shuffled, then the faces of the clip's last frame (which the
    # standalone clip's `extract` never yields) and of later shots with later track ids
    order = np.lexsort((rng.random(n), g["face_frame"]))
    extra = 40
    res = {"tracks": tracks + [[(20.0 + k / fr, (0.1, 0.1, 0.2, 0.2), "detection") for k in range(5)]],
           "face_T": np.concatenate([g["face_frame"][order] / fr, np.full(8, (int(g["video"][2]) - 1) / fr), 20.0 + np.arange(extra) / fr]),
           "face_id": np.concatenate([g["face_id"][order], np.arange(8), np.full(extra, len(tracks))]),
           "landmarks": np.concatenate([g["landmarks"][order].astype(np.int32), np.zeros((8 + extra, 68, 2), np.int32)]),
           "embeddings": np.concatenate([g["embeddings"][order], np.zeros((8 + extra, 128), np.float32)]),
           "labels": {0: 0}}
    c = golden.compare(g, res, prefix=True)
    assert c["all_exact"] and c["tracks"] == "exact" and c["face_rows"] == "exact" and c["landmarks"] == "exact" and "labels" not in c
    assert c["embed_l2_max"] == 0.0 and c["n_faces"] == n
    res["landmarks"][5, 3, 0] += 1
    bad = golden.compare(g, res, prefix=True)
    assert not bad["all_exact"] and bad["landmarks"] != "exact"
    res["landmarks"][5, 3, 0] -= 1
    res["tracks"][2] = res["tracks"][2][:-1]
    assert golden.compare(g, res, prefix=True)["tracks"] != "exact"
