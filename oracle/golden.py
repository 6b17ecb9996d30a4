"""ORACLE (test infrastructure): whole-clip fixtures of the CPU oracle flow on the benched configurations, and the comparison of a
product result with them.

A fixture (tests/golden/<name>.npz, written by tests/golden/make_full_clip.py HERE, where the oracle has the minutes a whole clip
costs it) freezes what `oracle.ref_flow` returns for EVERY frame of a synthetic clip that bench.py times:

  tracks     the rows of `pyannote-face.py track` (reference scripts/pyannote-face.py:239-268; tracking.py:331-357,374-434):
             frame index, track id, integer box (the normalised floats of the file are box / frame size), status string
  faces      the rows of `extract` (scripts/pyannote-face.py:121-175,271-314): frame index, track id, 68 integer points,
             the float32 descriptor as the embedder returned it (before the '%.5f' of the file)
  labels     cluster label per track (face/clustering.py:92-119,138-148)
  raw        per frame the detector's raw candidates BEFORE non-maximum suppression (level, filter, row, column, score bits):
             the screening pass of the product (csrc/screen.hip) decides which windows reach the exact chain, so the set of raw
             candidates -- not just the boxes that survive -- is what pins it

PARITY UNPINNED like the oracle itself (oracle/pvo.h): the fixture freezes the restated algorithms, not dlib's.

Only tests/, __graft_entry__.smoke() and bench.py's parity legs may import this module; the product never does.
"""
import hashlib
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# the clips bench.py times (same constructor arguments: bench.py `defaults`, `bench_farm`, `bench_stream`)
CLIPS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "c2_full": dict(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0),
    # BASELINE.json configs[3]: clip 0 of the 720p farm
    "c4_clip0": dict(width=1280, height=720, n_frames=250, n_shots=2, faces=8, seed=20260925, frame_rate=25.0),
    # BASELINE.json configs[2]: the first 1000-frame clip of the long streamed video (identities from a pool of 250: bench_stream)
    "c3_clip0": dict(width=1920, height=1080, n_frames=1000, n_shots=4, faces=8, seed=20260925, frame_rate=25.0, identities=250),
    # BASELINE.json configs[4]: the first shot (250 frames) of the 4K / 50 fps / 40 faces clip
    "c5_shot0": dict(width=3840, height=2160, n_frames=500, n_shots=2, faces=40, seed=20260925, frame_rate=50.0, take=250),
}


def video_of(name):
    """(SyntheticVideo, number of its leading frames the fixture covers)"""
    from pyannote_video_amd import synth
    va = dict(CLIPS[name])
    take = va.pop("take", None)
    v = synth.SyntheticVideo(**va)
    return v, (take if take is not None else v.n_frames)


def matches(name, **video_args):
    """does a bench video built with these SyntheticVideo arguments start with the clip fixture `name` was made from?"""
    want = {k: v for k, v in CLIPS[name].items() if k != "take"}
    want.setdefault("identities", 12)
    have = dict(video_args)
    have.setdefault("identities", 12)
    return want == have


def shots_of(v, take):
    """the video's shots that begin inside its first `take` frames"""
    return [(a, b) for a, b in v.shots() if a < take / v.frame_rate]


def path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def available(name):
    return os.path.exists(path(name))


def raw_key(level, filt, r, c, score_bits):
    """canonical order of one frame's raw candidates: by (level, filter, row, column) -- independent of the order either side found them in"""
    a = np.stack([np.asarray(level, np.int64), np.asarray(filt, np.int64), np.asarray(r, np.int64), np.asarray(c, np.int64),
                  np.asarray(score_bits, np.int64)], axis=1) if len(level) else np.zeros((0, 5), np.int64)
    order = np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0])) if len(a) else np.zeros(0, np.int64)
    return np.ascontiguousarray(a[order].astype(np.int32))


def raw_digest(rows):
    """8 bytes of sha256 over the canonical int32 [n, 5] table of a frame"""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(rows, np.int32).tobytes()).digest()[:8], np.uint64)[0]


def pack(video_args, tracks, lm_lines, em_rows, labels, raw_per_frame, frame_rate, size, seconds=None, threads=None):
    """the oracle flow's outputs -> the arrays of a fixture.  tracks: ref_flow.track_video's list; lm_lines: ref_flow.extract's landmark
    lines; em_rows: [(T, id, float32[128])] in the order extract emitted them; raw_per_frame: [int32 [n, 5]] canonical"""
    w, h = size
    t_rows, statuses = [], []
    for ident, tr in enumerate(tracks):
        for t, (l, tp, r, b), st in tr:
            if st not in statuses:
                statuses.append(st)
            box = [int(round(l * w)), int(round(tp * h)), int(round(r * w)), int(round(b * h))]
            # the file's floats are box / size: the fixture keeps the integers and the check re-divides
            assert (box[0] / w, box[1] / h, box[2] / w, box[3] / h) == (l, tp, r, b), "track box is not integer / size"
            t_rows.append([int(round(t * frame_rate)), ident] + box + [statuses.index(st)])
    t_rows = np.array(t_rows, np.int32).reshape(-1, 7)
    pts = np.array([[float(x) for x in line.split()[2:]] for line in lm_lines], np.float64).reshape(-1, 68, 2)
    pts = np.rint(pts * np.array([w, h], np.float64)).astype(np.int32)     # 5 decimals of x / width resolve the integer point
    assert np.abs(pts).max(initial=0) < 32768
    counts = np.array([len(r) for r in raw_per_frame], np.int32)
    return dict(video=np.array([video_args["width"], video_args["height"], video_args.get("take", video_args["n_frames"]), video_args["n_shots"],
                                video_args["faces"], video_args["seed"]], np.int64),
                frame_rate=np.float64(frame_rate),
                track_rows=t_rows, statuses=np.array(statuses),
                face_frame=np.array([int(round(T * frame_rate)) for T, _, _ in em_rows], np.int32),
                face_id=np.array([i for _, i, _ in em_rows], np.int32),
                landmarks=pts.astype(np.int16),
                embeddings=np.stack([e for _, _, e in em_rows]).astype(np.float32) if em_rows else np.zeros((0, 128), np.float32),
                labels=np.array(sorted((int(k), int(v)) for k, v in labels.items()), np.int32).reshape(-1, 2),
                raw_counts=counts, raw_rows=np.concatenate(raw_per_frame).astype(np.int32) if counts.sum() else np.zeros((0, 5), np.int32),
                raw_digest=np.array([raw_digest(r) for r in raw_per_frame], np.uint64),
                oracle_seconds=np.float64(seconds or 0.0), oracle_threads=np.int32(threads or 0))


def load(name):
    with np.load(path(name)) as z:
        g = {k: z[k] for k in z.files}
    g["name"] = name
    return g


def tracks_of(g):
    """the fixture's track rows in the shape ref_flow.track_video / FacePipeline.run return: [[(t, (l, t, r, b) / size, status)]]"""
    w, h = int(g["video"][0]), int(g["video"][1])
    fr = float(g["frame_rate"])
    st = [str(s) for s in g["statuses"]]
    out = [[] for _ in range(int(g["track_rows"][:, 1].max()) + 1)] if len(g["track_rows"]) else []
    for i, ident, l, tp, r, b, s in g["track_rows"].tolist():
        out[ident].append((i / fr, (l / w, tp / h, r / w, b / h), st[s]))
    return out


def prefix_of(g, res):
    """the part of a LONGER run's result that a fixture of the run's first shots covers: shots are tracked independently and their tracks
    come first, so the fixture's tracks are the run's first ones; the reference's `extract` never yields the LAST timestamp group of a
    file (pyannote-face.py:121-175), so the fixture holds the faces of every frame but its last one -- the longer run's faces before that
    frame are the ones to compare (labels are not: the longer run clusters more tracks)"""
    n_tr = int(g["track_rows"][:, 1].max()) + 1 if len(g["track_rows"]) else 0
    fr = float(g["frame_rate"])
    last = int(g["video"][2]) - 1
    ff = np.rint(np.asarray(res["face_T"], np.float64) * fr).astype(np.int64)
    fid = np.asarray(res["face_id"], np.int64)
    keep = np.nonzero((ff < last) & (fid < n_tr))[0]
    # the order of the faces of ONE timestamp is an artefact of pandas' unstable sort of the whole track table (formats.file_order), i.e.
    # of the table's size: both sides are put into (frame, track) order
    keep = keep[np.lexsort((fid[keep], ff[keep]))]
    return {"tracks": list(res["tracks"][:n_tr]), "face_T": np.asarray(res["face_T"])[keep], "face_id": fid[keep],
            "landmarks": np.asarray(res["landmarks"])[keep], "embeddings": np.asarray(res["embeddings"])[keep]}


def compare(g, res, labels=None, frame_offset=0, prefix=False):
    """product result (FacePipeline.run's dictionary; `labels` if the clustering ran outside it) against fixture g -> a dictionary of
    'exact' / first difference per output, made for one JSON line.  Every comparison is on the WHOLE clip (prefix: on the part of a longer
    run the fixture covers, see prefix_of; labels are then not compared)."""
    out = {"fixture": g["name"], "frames": int(g["video"][2])}
    if prefix:
        res = prefix_of(g, res)
        labels = None
        out["prefix_of_a_longer_run"] = True
        o = np.lexsort((g["face_id"], g["face_frame"]))
        g = dict(g, face_frame=g["face_frame"][o], face_id=g["face_id"][o], landmarks=g["landmarks"][o], embeddings=g["embeddings"][o])
    want = tracks_of(g)
    got = res["tracks"]
    if got == want:
        out["tracks"] = "exact"
    else:
        first = None
        for k, (a, b) in enumerate(zip(got, want)):
            if a != b:
                rows = [j for j, (x, y) in enumerate(zip(a, b)) if x != y]
                first = {"track": k, "row": rows[0] if rows else min(len(a), len(b)), "t": (a if rows else max(a, b, key=len))[rows[0] if rows else min(len(a), len(b)) - 1][0]}
                break
        out["tracks"] = {"MISMATCH": first or {"n_tracks": [len(got), len(want)]}}
    out["n_tracks"] = len(want)
    fr = float(g["frame_rate"])
    face_frame = np.rint(np.asarray(res["face_T"], np.float64) * fr).astype(np.int64) - frame_offset
    same_rows = len(face_frame) == len(g["face_frame"]) and np.array_equal(face_frame, g["face_frame"]) and \
        np.array_equal(np.asarray(res["face_id"], np.int64), g["face_id"].astype(np.int64))
    out["face_rows"] = "exact" if same_rows else "MISMATCH"
    out["n_faces"] = int(len(g["face_frame"]))
    if same_rows:
        lm = np.asarray(res["landmarks"]).astype(np.int64).reshape(-1, 68, 2)
        bad = np.nonzero((lm != g["landmarks"].astype(np.int64)).any(axis=(1, 2)))[0]
        out["landmarks"] = "exact" if len(bad) == 0 else {"MISMATCH": {"faces": int(len(bad)), "first_face_row": int(bad[0]), "frame": int(g["face_frame"][bad[0]])}}
        d = np.linalg.norm(np.asarray(res["embeddings"], np.float64) - g["embeddings"].astype(np.float64), axis=1)
        out["embed_l2_max"] = float(d.max(initial=0.0))
        out["embed_l2_bar"] = 1e-4
    else:
        out["landmarks"] = out["embed_l2_max"] = None
    lab = labels if labels is not None else (None if prefix else res.get("labels"))
    if lab is not None:
        mine = np.array(sorted((int(k), int(v)) for k, v in lab.items()), np.int32).reshape(-1, 2)
        out["labels"] = "exact" if np.array_equal(mine, g["labels"]) else "MISMATCH"
        out["n_clusters"] = int(len(set(g["labels"][:, 1].tolist())))
    out["all_exact"] = bool(out["tracks"] == "exact" and same_rows and out["landmarks"] == "exact" and out.get("labels", "exact") == "exact"
                            and out["embed_l2_max"] is not None and out["embed_l2_max"] <= 1e-4)
    return out


def compare_raw(g, raw_per_frame):
    """raw_per_frame: per frame the product's raw candidates as canonical int32 [n, 5] (raw_key) -> 'exact' or the first differing frame"""
    n = len(g["raw_counts"])
    if len(raw_per_frame) != n:
        return {"MISMATCH": {"frames": [len(raw_per_frame), n]}}
    off = np.concatenate([[0], np.cumsum(g["raw_counts"])])
    for i, rows in enumerate(raw_per_frame):
        if raw_digest(rows) != g["raw_digest"][i]:
            want = g["raw_rows"][off[i]:off[i + 1]]
            a = set(map(tuple, np.asarray(rows).tolist())); b = set(map(tuple, want.tolist()))
            return {"MISMATCH": {"first_frame": i, "product": len(rows), "oracle": int(len(want)), "only_product": sorted(a - b)[:4], "only_oracle": sorted(b - a)[:4]}}
    return "exact"
