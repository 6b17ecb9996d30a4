"""Face clustering -- drop-in for `pyannote.video.face.clustering.FaceClustering` (reference clustering.py:49-148).

Usage contract kept (clustering.py:130-134):
    >>> clustering = FaceClustering()
    >>> starting_point, features = clustering.model.preprocess(embedding)
    >>> result = clustering(starting_point, features=features)

The N x N `pdist` matrix (clustering.py:100-101), the T x T block means (:104-112) and the agglomeration loop of
pyannote.algorithms' HierarchicalAgglomerativeClustering run on the GPU (pvf_cluster_tracks)."""
import numpy as np
try:                                    # the reference's result types where they are installed (clustering.py:38,57,76) ...
    from pyannote.core import Segment, Annotation
except ImportError:                     # ... and stand-ins with the methods the face path touches where they are not
    from ._core import Segment, Annotation
try:                                    # the base class whose hooks pyannote.algorithms' agglomeration loop calls (clustering.py:40-43,49)
    from pyannote.algorithms.clustering.hac.model import HACModel as _HACModelBase
except ImportError:
    class _HACModelBase(object):
        """[EXT pyannote.algorithms HACModel, absent here] what the reference's `_Model` uses of it: the per-cluster models, kept by
        cluster name and read back with `model[cluster]` (clustering.py:90,107,109,117-118)"""

        def __init__(self, is_symmetric=False):
            self.is_symmetric = is_symmetric
            self._models = {}

        def __getitem__(self, cluster):
            return self._models[cluster]
try:
    from sortedcollections import ValueSortedDict
except ImportError:
    class ValueSortedDict(dict):
        """[EXT sortedcollections, absent here] a dict whose iteration order is by value; the two calls an agglomeration loop makes on
        the similarity matrix are kept: peekitem(-1) (the most similar pair) and plain item access / deletion"""

        def peekitem(self, index=-1):
            items = sorted(self.items(), key=lambda kv: kv[1])
            return items[index]

        def __iter__(self):
            return iter(k for k, _ in sorted(self.items(), key=lambda kv: kv[1]))
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

    def __getitem__(self, column):
        """`features['track']` / `features['time']` as the reference's DataFrame gives them (clustering.py:87)"""
        if column == "track":
            return self.track
        if column == "time":
            return self.time
        raise KeyError(column)


def _columns(features):
    """(track column, X) of what `preprocess` returned: this module's Features, or the reference's DataFrame (time, track, d0..d127)"""
    if isinstance(features, Features):
        return features.track, features.X
    return np.asarray(features["track"]), np.ascontiguousarray(np.asarray(features[features.columns[2:]]), np.float64)


class _Model(_HACModelBase):
    """Average Euclidean distance between face embeddings (reference _Model, clustering.py:49-119): the hook methods an
    agglomeration driver calls -- `compute_model`, `compute_merged_model`, `compute_similarity_matrix`, `compute_similarity` -- with the
    reference's arguments and return types.  The T x T block means come from the GPU (pvf_pair_mean_dist: K10) instead of an N x N
    `pdist` matrix on the host (clustering.py:100-101); a merged cluster's similarity is the size-weighted mean of its tracks' block
    means, which is the mean of the union block the reference takes (clustering.py:116-119)."""

    def __init__(self, ctx=None):
        super(_Model, self).__init__(is_symmetric=True)
        self.ctx = ctx

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

    # ---- the hooks (clustering.py:84-119) -------------------------------------------------------------------------------------------
    def compute_model(self, cluster, parent=None):
        """a cluster's model = the indices of its rows in parent.features (clustering.py:84-87)"""
        return np.where(_columns(parent.features)[0] == cluster)[0]

    def compute_merged_model(self, clusters, parent=None):
        """clustering.py:89-90 (a list, not a generator: numpy >= 2 refuses np.hstack of a generator -- SURVEY.md section 8 a9)"""
        return np.hstack([self[cluster] for cluster in clusters])

    def compute_similarity_matrix(self, parent=None):
        """{(cluster_i, cluster_j): - mean Euclidean distance of the |i| x |j| block}, both orders (clustering.py:92-114)"""
        clusters = list(self._models)
        _, X = _columns(parent.features)
        rows = [np.asarray(self[c], np.int64) for c in clusters]
        counts = np.array([len(r) for r in rows], np.int64)
        row_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        order = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        ctx = self.ctx or runtime.default_context()
        D = ctx.pair_mean_dist(np.ascontiguousarray(X[order], np.float64), row_start)
        # what compute_similarity needs later on: the block means, which initial cluster a row belongs to, the clusters' sizes
        self._block_mean = D
        self._cluster_of_row = np.full(len(X), -1, np.int64)
        self._cluster_of_row[order] = np.repeat(np.arange(len(clusters)), counts)
        matrix = ValueSortedDict()
        for i in range(len(clusters)):
            for j in range(i + 1, len(clusters)):
                similarity = -float(D[i, j])
                matrix[clusters[i], clusters[j]] = similarity
                matrix[clusters[j], clusters[i]] = similarity
        return matrix

    def compute_similarity(self, cluster1, cluster2, parent=None):
        """- mean distance between the rows of two (possibly merged) clusters (clustering.py:116-119)"""
        w1 = np.bincount(self._cluster_of_row[np.asarray(self[cluster1], np.int64)], minlength=len(self._block_mean)).astype(np.float64)
        w2 = np.bincount(self._cluster_of_row[np.asarray(self[cluster2], np.int64)], minlength=len(self._block_mean)).astype(np.float64)
        return -float(w1 @ self._block_mean @ w2) / (w1.sum() * w2.sum())


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
        self.model = _Model(ctx)
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
