"""Face clustering -- drop-in for `pyannote.video.face.clustering.FaceClustering` (reference clustering.py:49-148).

Usage contract kept (clustering.py:130-134):
    >>> clustering = FaceClustering()
    >>> starting_point, features = clustering.model.preprocess(embedding)
    >>> result = clustering(starting_point, features=features)

The N x N `pdist` matrix (clustering.py:100-101), the T x T block means (:104-112) and the agglomeration loop of
pyannote.algorithms' HierarchicalAgglomerativeClustering run on the GPU (pvf_cluster_tracks)."""
import numpy as np
from ._core import Segment, Annotation
from . import runtime
from . import formats


class Features(object):
    """what `preprocess` returns as `data` (a DataFrame in the reference): rows sorted by (track, time)"""

    def __init__(self, time, track, X):
        order = np.lexsort((time, track))
        if len(order) and np.array_equal(order, np.arange(len(order))):
            self.time, self.track, self.X = time, track, np.ascontiguousarray(X)          # already in (track, time) order: no copy
        else:
            self.time, self.track, self.X = time[order], track[order], np.ascontiguousarray(X[order])

    def __len__(self):
        return len(self.time)


class _Model(object):
    """Average Euclidean distance between face embeddings (reference _Model, clustering.py:49-119)"""

    def preprocess(self, embedding):
        """embedding: path of embedding.txt, or a (time, track, X) triple already in memory"""
        if isinstance(embedding, str):
            time, track, X = formats.read_embeddings(embedding)
        else:
            time, track, X = embedding
        data = Features(np.asarray(time, np.float64), np.asarray(track, np.int64), np.asarray(X, np.float64))
        starting_point = Annotation(modality='face')
        # rows are sorted by (track, time): a track's extent is its first and last row (the reference takes np.min / np.max of the
        # track's rows, clustering.py:76-77 -- the same two values, without a pass over all rows per track)
        tracks, first, count = np.unique(data.track, return_index=True, return_counts=True)
        t_min = data.time[first].tolist()
        t_max = data.time[first + count - 1].tolist()
        for trk, a, b in zip(tracks.tolist(), t_min, t_max):
            segment = Segment(a, b)
            if not segment:          # single-timestamp tracks are skipped (clustering.py:78-79)
                continue
            starting_point[segment, int(trk)] = int(trk)
        return starting_point, data


class FaceClustering(object):
    """Face clustering

    Parameters
    ----------
    threshold : float, optional    stop merging when the closest pair's mean distance exceeds it. Defaults to 0.6.
    force : bool, optional         passed to the stopping criterion like the reference does (clustering.py:138-141,
                                   DistanceThreshold(threshold=threshold, force=force)).  [EXT pyannote.algorithms]: with force the
                                   agglomeration runs on to a single cluster so that `history` holds the complete dendrogram, while
                                   the result is still the partition at which the threshold was crossed.
    metric : 'euclidean' (the reference, clustering.py:101) or 'cosine' (named by BASELINE.json's north_star; 1 - cos of the pair)
    """

    def __init__(self, threshold=0.6, force=False, logger=None, ctx=None, metric="euclidean"):
        if metric not in ("euclidean", "cosine"):
            raise ValueError("metric must be 'euclidean' or 'cosine'")
        self.force = bool(force)
        self.metric = metric
        self.threshold = threshold
        self.model = _Model()
        self.ctx = ctx
        self.logger = logger
        self.history = None
        self.shard = None      # dist.DistanceShard when the pairwise distances are split over the ranks of a job

    def cluster_arrays(self, track_ids, row_track, X):
        """labels for `track_ids` (sorted unique ints) given per-row track ids; X float64 [N, dim]"""
        ctx = self.ctx or runtime.default_context()
        order = np.argsort(row_track, kind="stable")
        keep = np.isin(row_track[order], track_ids)
        rows = order[keep]
        rt = row_track[rows]
        # (features that come from preprocess() are in (track, time) order already: when every track takes part nothing moves)
        Xs = np.ascontiguousarray(X, np.float64) if len(rows) == len(X) and np.array_equal(rows, np.arange(len(X))) else np.ascontiguousarray(X[rows], np.float64)
        counts = np.searchsorted(rt, track_ids, side="right") - np.searchsorted(rt, track_ids, side="left")     # rt is sorted (stable argsort above)
        row_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        cut = float("inf") if self.force else self.threshold
        if self.metric == "cosine":
            D = ctx.pair_mean_dist(Xs, row_start, metric=1)
            labels, log = ctx.cluster_dist(D, row_start, cut)
        elif self.shard is None:
            labels, log = ctx.cluster_tracks(Xs, row_start, cut)
        else:
            # several GPUs, every one holding all rows: each computes the distance-matrix rows of its share of the tracks
            # (balanced by row count), the rows are exchanged, and every rank agglomerates the same complete matrix
            t0, t1 = self.shard.track_range(row_start)
            U = self.shard.assemble(ctx.pair_upper_rows(Xs, row_start, t0, t1)[t0:t1], row_start)
            labels, log = ctx.cluster_upper(U, row_start, cut)
        self.history = [(int(track_ids[int(a)]), int(track_ids[int(b)]), float(d)) for a, b, d, _ in log]
        if self.force:
            # complete dendrogram in `history`; the partition returned is the one before the first merge above the threshold
            # (average linkage never merges at a smaller distance later on, so that is a prefix of the merge list)
            labels = np.arange(len(track_ids))
            for a, b, d, _ in log:
                if not (d <= self.threshold):
                    break
                labels[labels == int(b)] = int(a)
        return [int(track_ids[int(l)]) for l in labels]

    # ---- the in-memory path: float32 descriptors straight from the embedder, no float64 table on the host ---------------------------
    @staticmethod
    def plan_rows(time, track):
        """Index work of `preprocess` + `cluster_arrays` for rows given as (time, track) columns: -> (track ids that take part, sorted;
        order = the rows of those tracks in (track, time) order -- clustering.py:72 sort_values(by=['track', 'time']); row_start).
        Tracks whose extent is an empty segment (one timestamp) are left out like the reference leaves them out (clustering.py:76-79)."""
        time = np.asarray(time, np.float64)
        track = np.asarray(track, np.int64)
        order = np.lexsort((time, track))
        tr, tm = track[order], time[order]
        ids, first, count = np.unique(tr, return_index=True, return_counts=True)
        # Segment(t_min, t_max) is empty when its duration is not above pyannote.core's precision (1e-6): _core.Segment.__bool__
        keep = (tm[first + count - 1] - tm[first]) > 1e-6
        if not keep.all():
            order = order[np.repeat(keep, count)]
            ids, count = ids[keep], count[keep]
        row_start = np.concatenate([[0], np.cumsum(count)]).astype(np.int32)
        return ids, order.astype(np.int32), row_start

    def cluster_rows(self, time, track, emb, decimals=5, src_index=None):
        """{track: label} for float32 descriptor rows `emb` (numpy [n, 128] or runtime.DeviceRows) with their (time, track) columns (of
        row src_index[k] of `emb` when given): what
        preprocess((time, track, round(emb, 5))) followed by __call__ returns, with the rounding, the (track, time) gather, the pair
        means and the agglomeration on the device (pvf_cluster_tracks_f32; the split form over the ranks of a job when `shard` is set)."""
        ctx = self.ctx or runtime.default_context()
        ids, order, row_start = self.plan_rows(time, track)
        if len(ids) == 0:
            self.history = []
            return {}
        if src_index is not None:          # row k of (time, track) is emb[src_index[k]] (rows that were reordered without being moved)
            order = np.asarray(src_index)[order].astype(np.int32)
        cut = float("inf") if self.force else self.threshold
        if self.shard is None or self.metric == "cosine":
            labels, log = ctx.cluster_tracks_f32(emb, order, row_start, cut, decimals=decimals, metric=1 if self.metric == "cosine" else 0)
        else:
            labels, log = self.shard.cluster(ctx, emb, order, row_start, cut, decimals)
        return self._labels_of(ids, labels, log)

    def _labels_of(self, ids, labels, log):
        self.history = [(int(ids[int(a)]), int(ids[int(b)]), float(d)) for a, b, d, _ in log]
        if self.force:
            labels = np.arange(len(ids))
            for a, b, d, _ in log:
                if not (d <= self.threshold):
                    break
                labels[labels == int(b)] = int(a)
        return {int(t): int(ids[int(l)]) for t, l in zip(ids.tolist(), labels)}

    def __call__(self, starting_point, features=None):
        tracks = sorted(set(label for _, _, label in starting_point.itertracks(yield_label=True)))
        if not tracks:
            return starting_point.copy()
        labels = self.cluster_arrays(np.asarray(tracks, np.int64), features.track, features.X)
        return starting_point.rename_labels(dict(zip(tracks, labels)))
