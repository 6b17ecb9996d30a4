#!/usr/bin/env python
"""Generates tests/golden/reference_cli_small/ by running the REFERENCE'S OWN `track` and `extract`
(/root/reference/scripts/pyannote-face.py:239-314, executed verbatim through tests/refhost.py, with the reference's own
FaceTracking / TrackingByDetection / Face and the real munkres package) on a small synthetic clip, with `dlib` provided by the
CPU oracle (tests/oracle_dlib.py).  The three text files are what the reference writes: track.txt, landmarks.txt,
embedding.txt.  /root/reference exists in the build container only, so these files travel as fixtures; the GPU tests
compare the product's output with them byte for byte (tests/test_reference_binding.py).
    python tests/golden/make_reference_golden.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CLIP = dict(width=640, height=360, n_frames=12, n_shots=2, faces=3, min_face=50, max_face=110, seed=7)   # == conftest.small_video


def run_reference_cli(outdir, dlib_module, landmarks_path, embedding_path, clip=CLIP, every=0.0, record=None):
    """the reference's track() and extract() on the clip -> paths of the three files.  record: path of a dlib_trace.json to write
    (every call the reference makes into `dlib`, with the arguments and what came back: tests/dlib_trace.py)"""
    import refhost
    from pyannote_video_amd import synth
    video = synth.SyntheticVideo(**clip)
    if record:
        import dlib_trace
        dlib_module = dlib_trace.Recorder(dlib_module, [video.frame(i) for i in range(video.n_frames)])
    shot_path = os.path.join(outdir, "shots.json")
    with open(shot_path, "w") as f:
        json.dump(video.shots(), f)
    track_path, lm_path, emb_path = (os.path.join(outdir, n) for n in ("track.txt", "landmarks.txt", "embedding.txt"))
    with refhost.reference_modules(dlib_module, video_cls=synth.SyntheticVideo) as ref:
        ref.cli.track(video, shot_path, track_path, detect_every=every)           # CLI defaults: overlap 0.5, confidence 10, gap 1.0
        ref.cli.extract(video, landmarks_path, embedding_path, track_path, lm_path, emb_path)
    if record:
        dlib_module.save(record, {"clip": clip, "every": every, "what": "calls of /root/reference/scripts/pyannote-face.py track() + extract() into dlib "
                                                                          "(dlib = the CPU oracle), in order"})
    return track_path, lm_path, emb_path


def main():
    import oracle_dlib
    from pyannote_video_amd import models
    lp, ep = models.ensure_synthetic_models(tempfile.mkdtemp(), small=True)
    oracle_dlib.configure(models.load_container(models.DEFAULT_DETECTOR), models.dsst_tables())
    for name, every in (("reference_cli_small", 0.0), ("reference_cli_small_every3", 0.12)):     # --every=0.12 s = every 3rd frame
        out = os.path.join(HERE, name)
        os.makedirs(out, exist_ok=True)
        paths = run_reference_cli(out, oracle_dlib, lp, ep, every=every, record=os.path.join(out, "dlib_trace.json"))
        os.remove(os.path.join(out, "shots.json"))
        for p in paths:
            print(p, os.path.getsize(p), "bytes", sum(1 for _ in open(p)), "lines")


if __name__ == "__main__":
    main()
