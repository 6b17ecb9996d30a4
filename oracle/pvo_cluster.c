/*
 * pvo_cluster.c -- ORACLE (test infrastructure): face-track clustering.
 *   reference: pyannote/video/face/clustering.py:92-114 (-squareform(pdist(X,'euclidean')), block means),
 *              :116-119 (merged-pair similarity = mean of the union block), :138-141 (DistanceThreshold 0.6)
 * pdist is pinned against scipy in tests/; the HAC driver (pyannote.algorithms >= 0.8) is [EXT], PARITY UNPINNED
 * for tie order among equal similarities.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* clustering.py:104-112: the pairs i < j only (itertools.combinations), matrix[i, j] = matrix[j, i] = similarity; the diagonal stays 0
 * like squareform's.  (Up to round 3 this restatement summed both triangles independently -- D[j][i] could differ from D[i][j] in the
 * last bits, which the reference's mirrored matrix never does.) */
void pvo_pair_mean_dist(const double* X, int N, int dim, const int32_t* row_start, int T, double* D)
{
    (void)N;
    /* rows of the triangle are independent (each entry is one sequential chain whatever the thread count) */
    #pragma omp parallel for schedule(dynamic, 1) if (T > 8)
    for (int i = 0; i < T; ++i) {
        D[(size_t)i * T + i] = 0;
        for (int j = i + 1; j < T; ++j) {
            double sum = 0;
            for (int a = row_start[i]; a < row_start[i + 1]; ++a)
                for (int b = row_start[j]; b < row_start[j + 1]; ++b) {
                    double s = 0;
                    for (int k = 0; k < dim; ++k) { const double d = X[(size_t)a * dim + k] - X[(size_t)b * dim + k]; s += d * d; }
                    sum += sqrt(s);
                }
            const double cnt = (double)(row_start[i + 1] - row_start[i]) * (double)(row_start[j + 1] - row_start[j]);
            D[(size_t)i * T + j] = sum / cnt;
            D[(size_t)j * T + i] = sum / cnt;
        }
    }
}

/* mean over the union block == size-weighted mean of the two block means (clustering.py:116-119 recomputes it from
 * the raw block; equal in exact arithmetic, so labels -- not D -- are the parity object). */
int pvo_hac(const double* Din, const int32_t* sizes, int T, double threshold, int32_t* labels, double* merge_log)
{
    double* D = (double*)malloc(sizeof(double) * T * T);
    memcpy(D, Din, sizeof(double) * T * T);
    double* sz = (double*)malloc(sizeof(double) * T);
    char* alive = (char*)malloc(T);
    for (int i = 0; i < T; ++i) { sz[i] = sizes[i]; alive[i] = 1; labels[i] = i; }
    int merges = 0;
    for (;;) {
        int bi = -1, bj = -1; double bd = 0;
        for (int i = 0; i < T; ++i) {
            if (!alive[i]) continue;
            for (int j = i + 1; j < T; ++j) {
                if (!alive[j]) continue;
                if (bi < 0 || D[(size_t)i * T + j] < bd) { bd = D[(size_t)i * T + j]; bi = i; bj = j; }
            }
        }
        if (bi < 0 || bd > threshold) break;
        for (int k = 0; k < T; ++k) {
            if (!alive[k] || k == bi || k == bj) continue;
            const double v = (sz[bi] * D[(size_t)bi * T + k] + sz[bj] * D[(size_t)bj * T + k]) / (sz[bi] + sz[bj]);
            D[(size_t)bi * T + k] = v; D[(size_t)k * T + bi] = v;
        }
        sz[bi] += sz[bj]; alive[bj] = 0;
        for (int k = 0; k < T; ++k) if (labels[k] == bj) labels[k] = bi;
        if (merge_log) { merge_log[4 * merges] = bi; merge_log[4 * merges + 1] = bj; merge_log[4 * merges + 2] = bd; merge_log[4 * merges + 3] = sz[bi]; }
        ++merges;
    }
    free(D); free(sz); free(alive);
    return merges;
}
