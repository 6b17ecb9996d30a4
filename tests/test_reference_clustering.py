"""a7 / a9 pinned to the reference's own clustering.py, executed verbatim (its arithmetic is numpy / scipy / pandas, all installed):
`_Model.preprocess` (track extents, skipped empty segments, row order), `compute_model`, `compute_similarity_matrix` (minus the mean of the
|i| x |j| block of scipy's pdist) and `compute_similarity` of merged clusters, against the CPU oracle and the product's host code.
What stays unpinned is pyannote.algorithms' agglomeration loop (absent): see tests/refhost.reference_clustering_module."""
import os
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import refhost  # noqa: E402

pytestmark = pytest.mark.skipif(not refhost.have_reference(), reason="/root/reference is not present (GPU box)")


def _embedding_file(tmp_path, seed, n_tracks=14, n_ident=4):
    from pyannote_video_amd import formats
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 1, (n_ident, 128))
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    lines = []
    for trk in range(n_tracks):
        n = int(rng.integers(1, 9))
        t0 = float(rng.integers(0, 200)) / 25.0
        same_time = rng.random() < 0.15                        # a single-timestamp track: an empty segment, skipped by preprocess
        for k in range(n if not same_time else 1):
            x = centres[trk % n_ident] + 0.05 * rng.normal(0, 1, 128)
            x = 0.55 * x / np.linalg.norm(x)
            lines.append((t0 + k / 25.0, trk, x.astype(np.float32)))
    order = rng.permutation(len(lines))                        # the file is in time order in reality; any order must work
    path = str(tmp_path / "embedding.txt")
    with open(path, "w") as f:
        for i in order:
            T, trk, x = lines[i]
            f.write(formats.embedding_line(T, trk, x))
    return path


@pytest.mark.parametrize("seed", range(4))
def test_reference_model_similarities_equal_oracle_and_product_host(tmp_path, seed):
    from oracle import oracle
    from pyannote_video_amd import clustering as mine
    from pyannote_video_amd._core import Segment, Annotation
    path = _embedding_file(tmp_path, seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with refhost.reference_clustering_module(Segment, Annotation) as ref:
            model = ref._Model()
            starting_point, data = model.preprocess(path)

            class Parent(object):
                features = data
            tracks = sorted(set(int(t) for t in data["track"]))
            for trk in tracks:
                model._models[trk] = model.compute_model(trk, parent=Parent)
            matrix = model.compute_similarity_matrix(parent=Parent)
            merged = np.hstack([model[tracks[0]], model[tracks[1]]])          # what compute_merged_model means (np.hstack of a generator
            model._models["m"] = merged                                         # no longer runs on numpy >= 2: SURVEY.md section 8 a9)
            sim_merged = float(model.compute_similarity("m", tracks[2]))
            ref_rows = data[["time", "track"]].to_numpy()
            ref_X = np.array(data[data.columns[2:]])
            ref_start = sorted((s.start, s.end, int(t)) for s, t in starting_point.itertracks())
    # preprocess: same rows in the same (track, time) order, same starting point
    sp, feats = mine._Model().preprocess(path)
    assert np.array_equal(feats.time, ref_rows[:, 0]) and np.array_equal(feats.track, ref_rows[:, 1].astype(np.int64))
    assert np.array_equal(feats.X, ref_X)
    assert sorted((s.start, s.end, int(t)) for s, t in sp.itertracks()) == ref_start
    assert len(ref_start) < len(tracks) or all(np.sum(feats.track == t) > 1 for t in tracks)      # single-timestamp tracks were skipped
    # similarity matrix: - mean pairwise Euclidean distance, for every pair of tracks (including the skipped ones: clustering.py:100-112)
    row_start = np.concatenate([[0], np.cumsum([np.sum(feats.track == t) for t in tracks])]).astype(np.int32)
    D = oracle.pair_mean_dist(feats.X, row_start)
    for i, a in enumerate(tracks):
        for j, b in enumerate(tracks):
            if i != j:
                assert abs(-matrix[a, b] - D[i, j]) <= 1e-12 * max(1.0, D[i, j]), (a, b)
    # a merged cluster's similarity = size-weighted mean of its parts (what the GPU agglomeration updates with)
    n0, n1 = row_start[1] - row_start[0], row_start[2] - row_start[1]
    want = (n0 * D[0, 2] + n1 * D[1, 2]) / (n0 + n1)
    assert abs(-sim_merged - want) <= 1e-12


class _OracleContext(object):
    """stands where the GPU context stands on the CPU suite: the block means from the CPU oracle (checker, not product)"""

    def pair_mean_dist(self, X, row_start, metric=0):
        from oracle import oracle
        return oracle.pair_mean_dist(X, row_start)


@pytest.mark.parametrize("seed", range(3))
def test_product_hooks_drive_the_same_agglomeration_as_the_reference_hooks(tmp_path, seed):
    """seam S5: the reference's `_Model` (executed verbatim) and the product's, each driven through the SAME average-linkage loop over
    compute_model / compute_merged_model / compute_similarity_matrix / compute_similarity -- same models, same matrix (to 1e-12), same
    merges, same labels, same return types."""
    import hac_driver
    from pyannote_video_amd import clustering as mine
    from pyannote_video_amd._core import Segment, Annotation
    path = _embedding_file(tmp_path, 100 + seed, n_tracks=18, n_ident=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with refhost.reference_clustering_module(Segment, Annotation) as ref:
            model_r = ref._Model()
            sp_r, data_r = model_r.preprocess(path)
            tracks = sorted(int(t) for _, _, t in sp_r.itertracks(yield_label=True))
            first_model = model_r.compute_model(tracks[0], parent=hac_driver._Parent(data_r))
            labels_r, log_r = hac_driver.agglomerate(model_r, data_r, tracks, 0.6)
            matrix_type = type(ref.ValueSortedDict()).__mro__[1]
    model_p = mine._Model(ctx=_OracleContext())
    sp_p, data_p = model_p.preprocess(path)
    assert tracks == sorted(int(t) for _, _, t in sp_p.itertracks(yield_label=True))
    got = model_p.compute_model(tracks[0], parent=hac_driver._Parent(data_p))
    assert isinstance(got, np.ndarray) and np.array_equal(got, first_model)
    # the product's hooks accept the reference's DataFrame as parent.features too (a site that keeps its own preprocess)
    assert np.array_equal(mine._Model().compute_model(tracks[0], parent=hac_driver._Parent(data_r)), first_model)
    labels_p, log_p = hac_driver.agglomerate(model_p, data_p, tracks, 0.6)
    assert labels_p == labels_r
    assert [(a, b) for a, b, _ in log_p] == [(a, b) for a, b, _ in log_r] and len(log_r) > 0
    assert np.allclose([d for _, _, d in log_p], [d for _, _, d in log_r], rtol=0, atol=1e-12)
    m = mine._Model(ctx=_OracleContext())
    for c in tracks:
        m._models[c] = m.compute_model(c, parent=hac_driver._Parent(data_p))
    matrix = m.compute_similarity_matrix(parent=hac_driver._Parent(data_p))
    assert isinstance(matrix, dict) and isinstance(matrix, mine.ValueSortedDict) and matrix_type is dict
    assert set(matrix) == set((a, b) for a in tracks for b in tracks if a != b)
    assert isinstance(m.compute_similarity(tracks[0], tracks[1]), float)
    # the whole-replacement path agrees with the hook-driven one: FaceClustering.__call__ on the oracle-backed context's GPU twin is a
    # -m gpu test (tests/test_gpu_s5_hooks.py); here the oracle's own agglomeration
    from oracle import oracle
    rows = [np.where(data_p.track == t)[0] for t in tracks]
    row_start = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    D = oracle.pair_mean_dist(data_p.X[np.concatenate(rows)], row_start)
    lab, _ = oracle.hac(D, np.diff(row_start), 0.6)
    assert [tracks[int(l)] for l in lab] == [labels_p[t] for t in tracks]
