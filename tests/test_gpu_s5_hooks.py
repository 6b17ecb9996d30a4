"""Seam S5 on the GPU: the product's `_Model` hooks (reference clustering.py:84-119) driven by a plain average-linkage loop give the
labels of the whole-replacement path (`FaceClustering.__call__` -> pvf_cluster_tracks) and of the CPU oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


def _rows(seed, n_tracks=40, n_ident=7):
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 1, (n_ident, 128))
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    time, track, X = [], [], []
    for trk in range(n_tracks):
        n = int(rng.integers(2, 12))
        t0 = float(rng.integers(0, 500)) / 25.0
        for k in range(n):
            x = centres[trk % n_ident] + 0.05 * rng.normal(0, 1, 128)
            time.append(t0 + k / 25.0); track.append(trk); X.append(np.round(0.55 * x / np.linalg.norm(x), 5))
    return np.array(time), np.array(track, np.int64), np.array(X, np.float64)


@pytest.mark.parametrize("seed", range(3))
def test_hook_driven_agglomeration_equals_cluster_tracks_and_oracle(ctx, oracle, seed):
    import hac_driver
    from pyannote_video_amd import clustering
    time, track, X = _rows(seed)
    fc = clustering.FaceClustering(threshold=0.6, ctx=ctx)
    starting_point, features = fc.model.preprocess((time, track, X))
    tracks = sorted(int(t) for _, _, t in starting_point.itertracks(yield_label=True))
    # the hooks, one by one, through the driver
    labels, log = hac_driver.agglomerate(clustering._Model(ctx), features, tracks, 0.6)
    assert len(log) > 0 and len(set(labels.values())) < len(tracks)
    # the whole-replacement path
    result = fc(starting_point, features=features)
    whole = {int(t): int(l) for _, t, l in result.itertracks(yield_label=True)}
    assert whole == labels
    assert [(a, b) for a, b, _ in fc.history] == [(a, b) for a, b, _ in log]
    assert np.allclose([d for _, _, d in fc.history], [d for _, _, d in log], rtol=0, atol=1e-12)
    # the oracle
    rows = [np.where(features.track == t)[0] for t in tracks]
    row_start = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    D = oracle.pair_mean_dist(features.X[np.concatenate(rows)], row_start)
    lab, _ = oracle.hac(D, np.diff(row_start), 0.6)
    assert [tracks[int(l)] for l in lab] == [labels[t] for t in tracks]
    # the matrix hook alone against the oracle
    m = clustering._Model(ctx)
    for c in tracks:
        m._models[c] = m.compute_model(c, parent=hac_driver._Parent(features))
    matrix = m.compute_similarity_matrix(parent=hac_driver._Parent(features))
    for i, a in enumerate(tracks):
        for j, b in enumerate(tracks):
            if i != j:
                assert abs(-matrix[a, b] - D[i, j]) <= 1e-12 * max(1.0, D[i, j])
