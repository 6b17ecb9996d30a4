"""`pyannote-face` verbs of the product (pyannote_video_amd/cli.py; reference scripts/pyannote-face.py:29-89,239-314,415-455).
CPU: argument surface, video / shot readers.  GPU: track -> extract -> cluster on the small clip writes the files the reference's own
CLI wrote (tests/golden/reference_cli_small, see tests/test_reference_binding.py) and `identifier label` lines."""
import json
import os
import numpy as np
import pytest
from pyannote_video_amd import cli

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "reference_cli_small")
CLIP = dict(width=640, height=360, n_frames=12, n_shots=2, faces=3, min_face=50, max_face=110, seed=7)


def _lines(path):
    with open(path) as f:
        return f.read().splitlines()


def _canon(lines):
    return sorted(lines, key=lambda l: (float(l.split()[0]), int(l.split()[1])))


def test_video_and_shot_readers(tmp_path):
    frames = np.random.default_rng(0).integers(0, 255, (5, 8, 12, 3), dtype=np.uint8)
    p = str(tmp_path / "clip.npy")
    np.save(p, frames)
    v = cli.open_video(p, 50.0)
    assert v.size == (12, 8) and v.frame_size == (12, 8) and len(v) == 5 and v.frame_rate == 50.0
    got = list(v)
    assert [t for t, _ in got] == [0.0, 0.02, 0.04, 0.06, 0.08]
    assert all(f.flags["C_CONTIGUOUS"] and f.dtype == np.uint8 for _, f in got) and np.array_equal(got[3][1], frames[3])
    s = cli.open_video("synthetic:320x180x6:2:1:5", 25.0)
    assert s.size == (320, 180) and len(s) == 6 and s.n_shots == 2 and s.faces == 1 and s.seed == 5
    (tmp_path / "shots.json").write_text(json.dumps([[0, 1.5], [1.5, 4]]))
    assert [(x.start, x.end) for x in cli.load_shots(str(tmp_path / "shots.json"))] == [(0.0, 1.5), (1.5, 4.0)]
    (tmp_path / "tl.json").write_text(json.dumps({"pyannote": "Timeline", "content": [{"start": 0, "end": 2}]}))
    assert [(x.start, x.end) for x in cli.load_shots(str(tmp_path / "tl.json"))] == [(0.0, 2.0)]
    with pytest.raises(IOError):
        np.save(str(tmp_path / "bad.npy"), np.zeros((3, 4, 5), np.float32))
        cli.open_video(str(tmp_path / "bad.npy"), 25.0)


def test_argument_surface_matches_reference_options():
    """same verbs, positionals and track options as the reference's usage text (pyannote-face.py:36-65), same defaults (:112-114)"""
    seen = {}
    orig, orig_open = cli.track, cli.open_video
    try:
        cli.open_video = lambda spec, fps: spec
        cli.track = lambda video, shot, output, **kw: seen.update(kw, shot=shot, output=output)
        cli.main(["track", "synthetic:64x48x2", "shots.json", "out.txt"])
        assert seen == dict(detect_min_size=0.0, detect_every=0.0, track_min_overlap_ratio=0.5, track_min_confidence=10.0,
                            track_max_gap=1.0, ctx=None, shot="shots.json", output="out.txt")
        cli.main(["track", "--min-size=0.1", "--every=0.5", "--min-overlap=0.3", "--min-confidence=8", "--max-gap=0", "synthetic:64x48x2", "s", "o"])
        assert (seen["detect_min_size"], seen["detect_every"], seen["track_min_overlap_ratio"], seen["track_min_confidence"],
                seen["track_max_gap"]) == (0.1, 0.5, 0.3, 8.0, 0.0)
    finally:
        cli.track, cli.open_video = orig, orig_open


@pytest.mark.gpu
def test_track_extract_cluster_write_the_reference_files(tmp_path, ctx, model_paths):
    from oracle import ref_flow
    video = "synthetic:%dx%dx%d:%d:%d:%d" % (CLIP["width"], CLIP["height"], CLIP["n_frames"], CLIP["n_shots"], CLIP["faces"], CLIP["seed"])
    from pyannote_video_amd import synth
    assert cli.open_video(video, 25.0).size == (CLIP["width"], CLIP["height"])
    v = synth.SyntheticVideo(**CLIP)                     # the fixture clip also sets the generator's face-size arguments
    shots = str(tmp_path / "shots.json")
    with open(shots, "w") as f:
        json.dump(v.shots(), f)
    trk, lm, em, lab = (str(tmp_path / n) for n in ("track.txt", "landmarks.txt", "embedding.txt", "labels.txt"))
    cli.track(v, shots, trk, ctx=ctx)
    assert _lines(trk) == _lines(os.path.join(GOLD, "track.txt"))
    cli.extract(v, model_paths[0], model_paths[1], trk, lm, em, ctx=ctx)
    assert _canon(_lines(lm)) == _canon(_lines(os.path.join(GOLD, "landmarks.txt")))
    a = np.array([[float(x) for x in l.split()] for l in _canon(_lines(em))])
    b = np.array([[float(x) for x in l.split()] for l in _canon(_lines(os.path.join(GOLD, "embedding.txt")))])
    assert a.shape == b.shape and np.array_equal(a[:, :2], b[:, :2]) and np.linalg.norm(a[:, 2:] - b[:, 2:], axis=1).max() <= 1e-4 + 128 ** 0.5 * 1e-5
    labels = cli.cluster(em, lab, ctx=ctx)
    assert labels == ref_flow.cluster(_lines(os.path.join(GOLD, "embedding.txt")), 0.6)
    rows = [tuple(int(x) for x in l.split()) for l in _lines(lab)]
    assert [r[0] for r in rows] == sorted(set(int(l.split()[1]) for l in _lines(em)))
    assert all(labels.get(i, i) == l for i, l in rows)
    # force=True: complete dendrogram in the history, same partition
    from pyannote_video_amd.clustering import FaceClustering
    fc = FaceClustering(force=True, ctx=ctx)
    sp, feats = fc.model.preprocess(em)
    res = fc(sp, features=feats)
    assert {int(t): int(l) for _, t, l in res.itertracks(yield_label=True)} == labels
    assert len(fc.history) == len(labels) - 1
