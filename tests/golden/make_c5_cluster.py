"""Fixture of BASELINE.json configs[4]'s clustering stressor at full size (T = 10 000 tracks x 10 rows, N = 1e5: tools/c5_cluster.py's
generator and seed): the CPU oracle's block means (oracle/pvo_cluster.c: pvo_pair_mean_dist) agglomerated by an average-linkage loop that
is the oracle's pvo_hac statement for statement in its choices -- the first minimum in row-major order, the size-weighted update, labels =
the surviving smaller index -- but keeps a minimum per row, because pvo_hac's full scan per merge would take hours at this size.  The loop
is checked against oracle.hac itself (labels and every merge) at T = 1200 before the big run.  Minutes of CPU: frozen, not recomputed.
    python tests/golden/make_c5_cluster.py        -> tests/golden/c5_cluster_T10000.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pyannote-video_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def hac_rowmin(D, sizes, threshold):
    """oracle/pvo_cluster.c: pvo_hac with a cached minimum per row (first minimum of the row's j > i part; the global choice is the first
    row holding the global minimum): the same merges in the same order"""
    D = np.array(D, np.float64)
    T = D.shape[0]
    sz = np.asarray(sizes, np.float64).copy()
    alive = np.ones(T, bool)
    labels = np.arange(T)
    iu = np.triu(np.ones((T, T), bool), 1)
    W = np.where(iu, D, np.inf)                      # working copy: only j > i entries, dead rows / columns at +inf
    rmin = W.min(axis=1)
    rarg = W.argmin(axis=1)
    log = []
    while True:
        bi = int(np.argmin(rmin))
        bd = rmin[bi]
        if not np.isfinite(bd) or bd > threshold:
            break
        bj = int(rarg[bi])
        k = np.flatnonzero(alive)
        k = k[(k != bi) & (k != bj)]
        dbi = np.where(k < bi, W[k, bi], W[bi, k])   # D[bi][k] from the upper triangle
        dbj = np.where(k < bj, W[k, bj], W[bj, k])
        v = (sz[bi] * dbi + sz[bj] * dbj) / (sz[bi] + sz[bj])
        lo, hi = k[k < bi], k[k > bi]
        W[lo, bi] = v[k < bi]
        W[bi, hi] = v[k > bi]
        W[bj, :] = np.inf
        W[:, bj] = np.inf
        sz[bi] += sz[bj]
        alive[bj] = False
        labels[labels == bj] = bi
        log.append((bi, bj, bd, sz[bi]))
        # rows whose cached minimum may have changed: bi, bj, and the rows above them that pointed at bi / bj or got a smaller entry
        rmin[bj] = np.inf
        rmin[bi] = W[bi].min(); rarg[bi] = W[bi].argmin()
        redo = lo[(rarg[lo] == bi) | (rarg[lo] == bj)]
        up = k[k < bj]
        redo = np.union1d(redo, up[rarg[up] == bj])
        if len(redo):
            rmin[redo] = W[redo].min(axis=1); rarg[redo] = W[redo].argmin(axis=1)
        better = lo[(W[lo, bi] < rmin[lo]) | ((W[lo, bi] == rmin[lo]) & (bi < rarg[lo]))]
        rmin[better] = W[better, bi]; rarg[better] = bi
    return labels.astype(np.int32), np.array(log, np.float64).reshape(-1, 4)


def main():
    from oracle import oracle
    import c5_cluster
    # the loop against the oracle's own, where that one is affordable
    X, rs, _ = c5_cluster.make(1200, 3)
    D = oracle.pair_mean_dist(X, rs)
    la, loga = oracle.hac(D, np.diff(rs), 0.6)
    lb, logb = hac_rowmin(D, np.diff(rs), 0.6)
    assert np.array_equal(la, lb) and np.array_equal(np.asarray(loga)[:, :3], logb[:len(loga), :3]) and len(loga) == len(logb), "row-minimum loop differs from pvo_hac"
    print("row-minimum loop == oracle.hac at T = 1200: %d merges" % len(logb))
    t0 = time.time()
    X, rs, ident = c5_cluster.make(10000, 10)
    D = oracle.pair_mean_dist(X, rs)
    t1 = time.time()
    labels, log = hac_rowmin(D, np.diff(rs), 0.6)
    t2 = time.time()
    print("T = 10000, N = %d: block means %.0f s, agglomeration %.0f s, %d merges, %d clusters" % (len(X), t1 - t0, t2 - t1, len(log), len(set(labels.tolist()))))
    np.savez_compressed(os.path.join(HERE, "c5_cluster_T10000.npz"), labels=labels, merge_pairs=log[:, :2].astype(np.int32), merge_dist=log[:, 2],
                        T=10000, rows_per_track=10, seed=20260925, oracle_seconds=np.array([t1 - t0, t2 - t1]))


if __name__ == "__main__":
    main()
