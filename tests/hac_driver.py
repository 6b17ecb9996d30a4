"""A plain average-linkage loop over the four hook methods of an `HACModel` (reference clustering.py:84-119) -- what an agglomeration driver
such as pyannote.algorithms' HierarchicalAgglomerativeClustering does with them ([EXT], absent here): models per cluster, the similarity
matrix once, then merge the most similar pair, rebuild its model with compute_merged_model and its row of the matrix with
compute_similarity, until the best pair's mean distance exceeds the threshold (clustering.py:138-141).  Test infrastructure."""
import numpy as np


class _Parent(object):
    def __init__(self, features):
        self.features = features


def agglomerate(model, features, clusters, threshold):
    """-> ({cluster: surviving cluster}, [(kept, merged, mean distance)]).  Ties: the first pair in the order of `clusters` wins."""
    parent = _Parent(features)
    for c in clusters:
        model._models[c] = model.compute_model(c, parent=parent)
    sim = dict(model.compute_similarity_matrix(parent=parent))
    alive = list(clusters)
    label = {c: c for c in clusters}
    log = []
    while len(alive) > 1:
        best, pair = -np.inf, None
        for i, a in enumerate(alive):
            for b in alive[i + 1:]:
                if sim[a, b] > best:
                    best, pair = sim[a, b], (a, b)
        if -best > threshold:
            break
        a, b = pair
        try:
            merged = model.compute_merged_model([a, b], parent=parent)
        except TypeError:                      # the reference's np.hstack(generator) (clustering.py:90) on numpy >= 2
            merged = np.hstack([model[a], model[b]])
        model._models[a] = merged
        del model._models[b]
        alive.remove(b)
        for c in label:
            if label[c] == b:
                label[c] = a
        for c in alive:
            if c != a:
                sim[a, c] = sim[c, a] = model.compute_similarity(a, c, parent=parent)
        log.append((a, b, -float(best)))
    return label, log
