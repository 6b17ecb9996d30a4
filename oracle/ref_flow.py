"""ORACLE (test infrastructure): sequential restatement of the reference's host flow, one object and one frame at a time.

It follows the reference line by line in *behaviour* (not in code) and is deliberately the naive form: per-object trackers,
forward pass then backward pass, text-format round trips done literally with string formatting, clustering with scipy's
pdist and a plain Python agglomeration loop.  The product (pyannote-video_amd/) reaches the same results with batched,
lock-step GPU calls; tests compare the two.

  track_shot / track_video   pyannote/video/tracking.py:184-259 (_track), :261-357, :374-434
  extract                    scripts/pyannote-face.py:121-175 (getFaceGenerator), :271-314
  cluster                    pyannote/video/face/clustering.py:59-119,138-148 + pyannote.algorithms HAC ([EXT], PARITY UNPINNED)
"""
import itertools
import numpy as np
import networkx as nx
from scipy.spatial.distance import pdist, squareform
from . import oracle as O

DET, FWD, BWD = 'detection', 'forward', 'backward'


class DRect(object):
    """dlib.drectangle semantics needed by tracking.py:129-134"""

    def __init__(self, l, t, r, b):
        self.l, self.t, self.r, self.b = float(l), float(t), float(r), float(b)

    def area(self):
        if self.l > self.r or self.t > self.b:
            return 0.0
        return (self.r - self.l) * (self.b - self.t)

    def intersect(self, o):
        return DRect(max(self.l, o.l), max(self.t, o.t), min(self.r, o.r), min(self.b, o.b))


def match(r1, r2, ratio):
    ov = r1.intersect(r2).area()
    if ov < ratio * r1.area() or ov < ratio * r2.area():
        ov = 0.
    return ov


def associate(trackers, detections, ratio):
    """trackers: dict id -> object with get_position() -> (l,t,r,b)"""
    nt, nd = len(trackers), len(detections)
    if nt < 1 or nd < 1:
        return {}
    n = max(nt, nd)
    area = np.zeros((n, n))
    items = list(trackers.items())
    for ti, (_, trk) in enumerate(items):
        p = DRect(*trk.get_position())
        for di, det in enumerate(detections):
            area[ti, di] = match(p, DRect(*det), ratio)
    out = {}
    for ti, di in O.munkres(np.max(area) - area):
        if ti >= nt or di >= nd:
            continue
        if area[ti, di] > 0.:
            out[di] = items[ti][0]
    return out


def one_pass(graph, cache, direction, tracker_factory, min_conf, ratio, pool=None):
    """pool: optional concurrent.futures executor -- the trackers of one frame are independent objects, so the all-core CPU
    baseline updates / starts them concurrently (the C calls release the interpreter lock); results are identical"""
    seq = cache if direction == FWD else list(reversed(cache))
    trackers, conf, prev = {}, {}, {}
    next_id = 0
    for t, frame in seq:
        items = list(trackers.items())
        confs = list(pool.map(lambda it: it[1].update(frame), items)) if pool is not None else [trk.update(frame) for _, trk in items]
        for (ident, trk), c in zip(items, confs):
            conf[ident] = c
            if c < min_conf:
                del trackers[ident], conf[ident], prev[ident]
        detections = [d for _, d, status in graph[t] if status == DET]
        m = associate(trackers, detections, ratio)
        for di, ident in m.items():
            graph.add_edge(prev[ident], (t, detections[di], DET), confidence=conf[ident])
            del trackers[ident], conf[ident], prev[ident]
        for ident, trk in trackers.items():
            node = (t, tuple(trk.get_position()), direction)
            graph.add_edge(prev[ident], node, confidence=conf[ident])
            prev[ident] = node
        def start(det):
            trk = tracker_factory()
            trk.start_track(frame, tuple(float(v) for v in det))
            return trk
        for det, trk in zip(detections, list(pool.map(start, detections)) if pool is not None else [start(d) for d in detections]):
            trackers[next_id] = trk
            prev[next_id] = (t, det, DET)
            next_id += 1


def fix(track, ratio):
    out = []
    order = {DET: 2, FWD: 1, BWD: 3}
    for t, group in itertools.groupby(sorted(track), key=lambda x: x[0]):
        group = list(group)
        bad = any(match(DRect(*a[1]), DRect(*b[1]), ratio) == 0 for a, b in itertools.combinations(group, 2))
        status = "+".join(sorted((g[2] for g in group), key=lambda s: order[s]))
        if bad:
            status = "error(%s)" % status
        pos = tuple(int(round(v)) for v in np.mean(np.vstack([g[1] for g in group]), axis=0))
        out.append((t, pos, status))
    return out


def span(track):
    ts = [t for t, _, _ in track]
    return (min(ts), max(ts))


def fill_gaps(tracks, max_gap, ratio):
    tracks = sorted(tracks, key=span)
    g = nx.Graph()
    g.add_nodes_from(range(len(tracks)))
    for i, j in itertools.combinations(range(len(tracks)), 2):
        ti, tj = tracks[i][-1][0], tracks[j][0][0]
        if tj < ti or tj - ti > max_gap:
            continue
        if match(DRect(*tracks[i][-1][1]), DRect(*tracks[j][0][1]), ratio):
            g.add_edge(i, j)
    return [[item for k in sorted(comp) for item in tracks[k]] for comp in nx.connected_components(g)]


def track_shot(cache, detections, tracker_factory, min_conf=10., ratio=0.3, max_gap=0., pool=None):
    """cache [(t, frame)], detections [[box]] aligned with cache -> list of tracks sorted by (min_t, max_t)"""
    graph = nx.DiGraph()
    for (t, _), dets in zip(cache, detections):
        graph.add_node(t)
        for d in dets:
            graph.add_edge(t, (t, tuple(d), DET))
    one_pass(graph, cache, FWD, tracker_factory, min_conf, ratio, pool)
    one_pass(graph, cache, BWD, tracker_factory, min_conf, ratio, pool)
    graph.remove_nodes_from([n for n in list(graph) if not isinstance(n, tuple)])
    comps = nx.connected_components(graph.to_undirected(reciprocal=False))
    tracks = fill_gaps([fix(c, ratio) for c in comps], max_gap, ratio)
    return sorted(tracks, key=span)


def track_video(frames, times, shots, detect, tracker_factory, frame_rate, detect_every=0., min_conf=10., ratio=0.3, max_gap=0., pool=None):
    """-> normalised tracks in the order `pyannote-face.py track` enumerates them"""
    every = int(detect_every * frame_rate) if detect_every > 0 else 1
    every = max(every, 1)
    h, w = frames[0].shape[:2]
    ends = [s[1] for s in shots]
    out, cache, dets, k = [], [], [], 0

    def flush():
        for tr in track_shot(cache, dets, tracker_factory, min_conf, ratio, max_gap, pool):
            out.append([(t, (l / w, tp / h, r / w, b / h), st) for t, (l, tp, r, b), st in tr])
    for i, (t, frame) in enumerate(zip(times, frames)):
        # time-driven segment generator: once t reaches the current segment's end, flush and move to the next segment
        if k < len(ends) and not (ends[k] > t):
            flush()
            cache, dets = [], []
            k += 1
        cache.append((t, frame))
        dets.append([tuple(d) for d in detect(frame)] if i % every == 0 else [])
    flush()
    return out


def track_text(tracks):
    lines = []
    for ident, track in enumerate(tracks):
        for t, (l, tp, r, b), status in track:
            lines.append('%.3f %d %.3f %.3f %.3f %.3f %s' % (t, ident, l, tp, r, b, status))
    return lines


def extract(track_lines, frames, times, landmarks, embed, pool=None, keep=None):
    """-> (landmark lines, embedding lines), literal text like the CLI writes.  pool: optional executor, faces of one frame are
    independent (all-core CPU baseline).  keep: optional list that receives (T, track, float32 descriptor) per face, the values
    before the '%.5f' of the file (whole-clip fixtures, oracle/golden.py)"""
    h, w = frames[0].shape[:2]
    rows = []
    for line in track_lines:
        p = line.split()
        rows.append((float(p[0]), int(p[1]), [np.float32(v) for v in p[2:6]]))
    # tracking.sort_values('t') (pyannote-face.py:130): pandas' default sort is numpy's unstable quicksort argsort
    rows = [rows[i] for i in np.argsort(np.array([r[0] for r in rows], np.float64), kind="quicksort")]

    def generator():
        t = yield
        faces, current = [], None
        for T, ident, (l, tp, r, b) in rows:
            # iterrows() over the mixed-dtype frame yields Python floats: the products are float64 (pyannote-face.py:142-145)
            box = (int(float(l) * w), int(float(tp) * h), int(float(r) * w), int(float(b) * h))
            if T == current or current is None:
                faces.append((ident, box))
                current = T
                continue
            while True:
                if current > t:
                    t = yield t, []
                    continue
                t = yield current, faces
                faces, current = [(ident, box)], T
                break
        while True:
            t = yield t, []
    gen = generator()
    gen.send(None)
    lm_lines, em_lines = [], []
    for t, frame in zip(times, frames):
        T, faces = gen.send(t)
        def one(face):
            pts = landmarks(frame, face[1])
            return pts, embed(frame, pts)
        for (ident, box), (pts, e) in zip(faces, list(pool.map(one, faces)) if pool is not None else [one(f) for f in faces]):
            lm_lines.append('%.3f %d' % (T, ident) + ''.join(' %.5f %.5f' % (x / w, y / h) for x, y in pts))
            em_lines.append('%.3f %d' % (T, ident) + ''.join(' %.5f' % float(v) for v in e))
            if keep is not None:
                keep.append((T, ident, np.array(e, np.float32)))
    return lm_lines, em_lines


def cluster(embedding_lines, threshold=0.6):
    """-> {track: label} for the tracks that take part (non-empty extent)"""
    data = np.array([[float(v) for v in line.split()] for line in embedding_lines], np.float64).reshape(-1, 130)
    order = np.lexsort((data[:, 0], data[:, 1]))
    data = data[order]
    time, track, X = data[:, 0], data[:, 1].astype(int), data[:, 2:]
    names = [int(k) for k in np.unique(track) if (time[track == k].max() - time[track == k].min()) > 1e-6]
    if not names:
        return {}
    neg = -squareform(pdist(X, metric='euclidean'))
    model = {k: np.where(track == k)[0] for k in names}
    sim = {}
    for a, b in itertools.combinations(names, 2):
        sim[a, b] = np.mean(neg[model[a]][:, model[b]])
    label = {k: k for k in names}
    while len(model) > 1:
        (a, b), s = max(sim.items(), key=lambda kv: (kv[1], -kv[0][0], -kv[0][1]))
        if s < -threshold:
            break
        model[a] = np.hstack([model[a], model[b]])
        del model[b]
        for k in label:
            if label[k] == b:
                label[k] = a
        sim = {p: v for p, v in sim.items() if b not in p}
        for c in model:
            if c == a:
                continue
            key = (min(a, c), max(a, c))
            sim[key] = np.mean(neg[model[a]][:, model[c]])
    return label
