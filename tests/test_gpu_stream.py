"""GPU: the streaming engine on the real library.
  * FacePipeline.run_stream (numpy frames through the pinned ingest ring, frames released shot by shot) == FacePipeline.run (frames
    resident) bit for bit -- tracks, landmarks, embeddings, labels -- also with windowed bulk tracker starts and with --min-size;
  * FacePipeline.run_many (several clips, one engine run) == one run per clip;
  * the `process` verb (one pass) writes the files of `track` + `extract`, and those are the reference CLI's own files (tests/golden);
  * end-to-end parity against the CPU oracle flow at the shapes of BASELINE.json configs[3] (720p clip), configs[4] (4K, 50 fps, 40
    faces, windowed trackers) and configs[1] with `--every 0.5` (trackers that live for 12 frames, reference tracking.py:383-386,425)."""
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _same(a, b):
    assert a["tracks"] == b["tracks"]
    assert a["face_T"].tolist() == b["face_T"].tolist() and a["face_id"].tolist() == b["face_id"].tolist()
    assert a["face_boxes"] == b["face_boxes"]
    assert np.array_equal(a["landmarks"], b["landmarks"]) and np.array_equal(a["embeddings"], b["embeddings"])
    assert a["labels"] == b["labels"]


@pytest.fixture(scope="module")
def clip6():
    from pyannote_video_amd import synth
    return synth.SyntheticVideo(width=640, height=360, n_frames=36, n_shots=6, faces=3, min_face=50, max_face=110, seed=17)


def test_streamed_run_equals_resident_run(ctx, clip6, model_paths):
    from pyannote_video_amd import pipeline
    v = clip6
    frames_np = [v.frame(i) for i in range(v.n_frames)]
    times = [v.timestamp(i) for i in range(v.n_frames)]
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_batch_size=4)
    base = pipe.run([ctx.upload(f) for f in frames_np], times, v.frame_rate, v.shots())
    assert len(base["tracks"]) >= 12 and len(base["labels"]) >= 6
    free0, _ = ctx.mem_info()
    got = pipe.run_stream(v, v.shots())                     # the video object itself: iterated once, numpy frames
    _same(got, base)
    assert got["frames"] == v.n_frames and got["peak_frames_resident"] < v.n_frames
    # windowed bulk tracker starts (every shot over the limit, windows of 4 detections), pipelined and sequential
    for overlap in (True, False):
        p2 = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_batch_size=4, overlap=overlap, speculate_limit=0, speculate_window=4)
        got = p2.run_stream(list(zip(times, frames_np)), v.shots(), frame_rate=v.frame_rate, size=v.size)
        _same(got, base)
        assert p2.last_engine.stats["windowed_shots"] == 6
    # track-only (the `track` verb's mode): same tracks, handed over shot by shot
    seen = []
    t_only = pipe.run_stream(v, v.shots(), extract=False, on_tracks=seen.append)
    assert t_only["tracks"] == base["tracks"] and [t for s in seen for t in s] == base["tracks"] and len(seen) == 6
    # the buffers of the released frames are in the pool, and trimming it gives them back
    assert ctx.pool_trim(0) == 0
    free1, _ = ctx.mem_info()
    assert free1 >= free0 - (64 << 20)


def test_streamed_min_size_equals_resident(ctx, clip6, model_paths):
    from pyannote_video_amd import pipeline
    v = clip6
    frames_np = [v.frame(i) for i in range(v.n_frames)]
    times = [v.timestamp(i) for i in range(v.n_frames)]
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_min_size=0.14, detect_batch_size=4)
    base = pipe.run([ctx.upload(f) for f in frames_np], times, v.frame_rate, v.shots())
    assert len(base["tracks"]) >= 6
    _same(pipe.run_stream(v, v.shots()), base)


def test_many_clips_in_one_engine_run_equal_one_run_each(ctx, model_paths):
    from pyannote_video_amd import synth, pipeline
    pipe = pipeline.FacePipeline(ctx, model_paths[0], model_paths[1], detect_batch_size=4)
    vids = [synth.SyntheticVideo(width=640, height=360, n_frames=10, n_shots=2, faces=2 + (k % 2), min_face=50, max_face=110, seed=40 + k) for k in range(4)]
    clips, singles = [], []
    for v in vids:
        fr = [ctx.upload(v.frame(i)) for i in range(v.n_frames)]
        t = [v.timestamp(i) for i in range(v.n_frames)]
        clips.append(dict(frames=fr, times=t, frame_rate=v.frame_rate, shots=v.shots()))
        singles.append(pipe.run(fr, t, v.frame_rate, v.shots()))
    order = []
    res = pipe.run_many(clips, on_result=lambda k, r: order.append(k))
    assert order == [0, 1, 2, 3]
    for a, b in zip(res, singles):
        _same(a, b)
    # streamed clips (video objects) through the same call
    res2 = pipe.run_many([dict(video=v, shots=v.shots()) for v in vids])
    for a, b in zip(res2, singles):
        _same(a, b)


def test_process_verb_writes_the_files_of_track_plus_extract(tmp_path, ctx, model_paths):
    from pyannote_video_amd import cli, synth
    from tests.test_cli import CLIP, GOLD, _lines
    v = synth.SyntheticVideo(**CLIP)
    shots = str(tmp_path / "shots.json")
    with open(shots, "w") as f:
        json.dump(v.shots(), f)
    p = {k: str(tmp_path / k) for k in ("t1", "l1", "e1", "lab1", "t2", "l2", "e2", "lab2")}
    cli.process(v, shots, model_paths[0], model_paths[1], p["t1"], p["l1"], p["e1"], p["lab1"], ctx=ctx)
    cli.track(v, shots, p["t2"], ctx=ctx)
    cli.extract(v, model_paths[0], model_paths[1], p["t2"], p["l2"], p["e2"], ctx=ctx)
    cli.cluster(p["e2"], p["lab2"], ctx=ctx)
    for a, b in (("t1", "t2"), ("l1", "l2"), ("e1", "e2"), ("lab1", "lab2")):
        assert _lines(p[a]) == _lines(p[b]), a                       # line for line, order included
    assert _lines(p["t1"]) == _lines(os.path.join(GOLD, "track.txt"))          # the file the reference's own CLI wrote
    assert sorted(_lines(p["l1"])) == sorted(_lines(os.path.join(GOLD, "landmarks.txt")))


def _oracle_flow(oracle, frames_np, times, shots, fps, model_paths, every=0.0, threads=64):
    from pyannote_video_amd import models
    from oracle import ref_flow
    import concurrent.futures
    oracle.lib().pvo_set_threads(min(threads, oracle.usable_cpus(cap=1024)))
    det = oracle.Detector(models.load_container(models.DEFAULT_DETECTOR))
    sp = oracle.ShapePredictor(models.load_container(model_paths[0]))
    emb = oracle.Embedder(models.load_container(model_paths[1]))
    tabs = models.dsst_tables()
    pool = concurrent.futures.ThreadPoolExecutor(16)
    tracks = ref_flow.track_video(frames_np, times, shots, det, lambda: oracle.Tracker(tabs), fps, detect_every=every, min_conf=10., ratio=0.5, max_gap=1.0, pool=pool)
    lm, em = ref_flow.extract(ref_flow.track_text(tracks), frames_np, times, sp, emb, pool=pool)
    pool.shutdown()
    # the landmark lines hold x / width, y / height with 5 decimals: that resolves the integer point (the faces of a frame run in a
    # thread pool, so the lines -- written in order -- are the record, not the order of the calls)
    h, w = frames_np[0].shape[:2]
    pts = np.rint(np.array([[float(x) for x in l.split()[2:]] for l in lm]).reshape(-1, 68, 2) * np.array([w, h], np.float64)).astype(np.int64)
    return tracks, pts, em, ref_flow.cluster(em, 0.6)


def _check_against_oracle(res, ref):
    tracks, pts, em, labels = ref
    assert res["tracks"] == tracks
    assert len(pts) == len(res["landmarks"]) > 0 and np.array_equal(pts, res["landmarks"].astype(np.int64))
    ref_e = np.array([[float(x) for x in line.split()[2:]] for line in em]).reshape(-1, 128)
    assert [int(l.split()[1]) for l in em] == res["face_id"].tolist()
    assert np.linalg.norm(ref_e - res["embeddings"], axis=1).max() <= 1e-4 + 128 ** 0.5 * 5e-6        # bar 1e-4 + the text rounding of the reference side
    assert res["labels"] == labels


def test_gpu_parity_c4_clip(ctx_full, oracle, full_model_paths):
    """configs[3] shape: a 720p clip, 8 faces, a window around its shot cut, the whole flow against the CPU oracle (FULL landmark model,
    like every oracle-checked end-to-end window since round 4)"""
    from pyannote_video_amd import synth, pipeline
    v = synth.SyntheticVideo(width=1280, height=720, n_frames=250, n_shots=2, faces=8, seed=20260925)
    idx = list(range(121, 131))
    frames_np = [v.frame(i) for i in idx]
    times = [v.timestamp(i) for i in idx]
    pipe = pipeline.FacePipeline(ctx_full, full_model_paths[0], full_model_paths[1], detect_batch_size=8)
    res = pipe.run_many([dict(frames=[ctx_full.upload(f) for f in frames_np], times=times, frame_rate=v.frame_rate, shots=v.shots())])[0]
    assert len(res["tracks"]) >= 12
    _check_against_oracle(res, _oracle_flow(oracle, frames_np, times, v.shots(), v.frame_rate, full_model_paths))


def test_gpu_parity_c5_e2e(ctx_full, oracle, full_model_paths):
    """configs[4] shape: 3840 x 2160, 50 fps, 40 faces per frame, six frames across a cut, bulk tracker starts in windows of 64 (every shot is
    over the limit), streamed through the ingest ring: tracks, landmarks, embeddings and labels against the CPU oracle"""
    from pyannote_video_amd import synth, pipeline
    v = synth.SyntheticVideo(width=3840, height=2160, n_frames=500, n_shots=2, faces=40, seed=20260925, frame_rate=50.0)
    idx = list(range(247, 253))
    frames_np = [v.frame(i) for i in idx]
    times = [v.timestamp(i) for i in idx]
    pipe = pipeline.FacePipeline(ctx_full, full_model_paths[0], full_model_paths[1], detect_batch_size=4, speculate_limit=100, speculate_window=64)
    res = pipe.run_stream(list(zip(times, frames_np)), v.shots(), frame_rate=v.frame_rate, size=v.size)
    assert pipe.last_engine.stats["windowed_shots"] == 2 and len(res["tracks"]) >= 60
    _check_against_oracle(res, _oracle_flow(oracle, frames_np, times, v.shots(), v.frame_rate, full_model_paths))


def test_gpu_parity_1080p_every_half_second(ctx_full, oracle, full_model_paths):
    """`--every 0.5` at the benched size (reference tracking.py:383-386,425): detection on every 12th frame, trackers carry the faces in
    between (committed deferred updates, on-demand updates frame after frame, both passes)"""
    from pyannote_video_amd import synth, pipeline
    v = synth.SyntheticVideo(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925)
    idx = list(range(236, 264))                               # frames 240 and 252 are detection frames; the cut is at 250
    frames_np = [v.frame(i) for i in idx]
    times = [v.timestamp(i) for i in idx]
    # frame indices count from the start of the video: hand the window over with its own numbering (i % every on 236..263 differs from
    # 0..27), i.e. run the reference flow and the product on a clip that STARTS at frame 236: both count from 0
    pipe = pipeline.FacePipeline(ctx_full, full_model_paths[0], full_model_paths[1], detect_every=0.5, detect_batch_size=8)
    res = pipe.run([ctx_full.upload(f) for f in frames_np], times, v.frame_rate, v.shots())
    ref = _oracle_flow(oracle, frames_np, times, v.shots(), v.frame_rate, full_model_paths, every=0.5)
    assert any("forward" in st and "backward" in st for tr in ref[0] for _, _, st in tr)
    _check_against_oracle(res, ref)
