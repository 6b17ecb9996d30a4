/*
 * pvo_ert.c -- ORACLE (test infrastructure): dlib shape_predictor (ensemble of regression trees).
 *   reference: pyannote/video/face/face.py:58,69-70 ; caller scripts/pyannote-face.py:296
 * PARITY UNPINNED ([EXT] restatement of dlib/image_processing/shape_predictor.h).
 *
 * Orders the HIP kernel reproduces: similarity fit sums over parts i = 0..P-1 sequentially in double;
 * leaf accumulation per coordinate over trees 0..T-1 sequentially in float.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* linear part of find_similarity_transform(from -> to) [EXT dlib/geometry/point_transforms.h, Umeyama];
 * for proper rotations the optimum is the complex least-squares ratio  sum(conj(f) t) / sum |f|^2 . */
static void similarity_linear(const float* from, const float* to, int n, float M[4])
{
    double mfx = 0, mfy = 0, mtx = 0, mty = 0;
    for (int i = 0; i < n; ++i) { mfx += from[2 * i]; mfy += from[2 * i + 1]; mtx += to[2 * i]; mty += to[2 * i + 1]; }
    mfx /= n; mfy /= n; mtx /= n; mty /= n;
    double a = 0, b = 0, s = 0;
    for (int i = 0; i < n; ++i) {
        const double fx = from[2 * i] - mfx, fy = from[2 * i + 1] - mfy;
        const double tx = to[2 * i] - mtx, ty = to[2 * i + 1] - mty;
        a += fx * tx + fy * ty;
        b += fx * ty - fy * tx;
        s += fx * fx + fy * fy;
    }
    const double ca = a / s, cb = b / s;
    M[0] = (float)ca; M[1] = (float)(-cb);
    M[2] = (float)cb; M[3] = (float)ca;
}

void pvo_landmarks(const uint8_t* rgb, int h, int w, const int32_t rect[4], const pvo_shape_model* m, int32_t* pts)
{
    const int P = m->n_parts, NP = m->n_pix, T = m->n_trees;
    const int n_split = (1 << m->depth) - 1, n_leaf = 1 << m->depth;
    float* cur = (float*)malloc(sizeof(float) * 2 * P);
    float* pix = (float*)malloc(sizeof(float) * NP);
    memcpy(cur, m->initial_shape, sizeof(float) * 2 * P);
    /* unnormalizing_tform(rect): (0,0)->tl, (1,0)->tr, (1,1)->br  =>  x = l + u*(r-l), y = t + v*(b-t) */
    const double sx = (double)rect[2] - (double)rect[0], sy = (double)rect[3] - (double)rect[1];
    const double ox = rect[0], oy = rect[1];
    for (int it = 0; it < m->n_cascades; ++it) {
        float M[4];
        similarity_linear(m->initial_shape, cur, P, M);
        const int32_t* anchor = m->anchor_idx + (size_t)it * NP;
        const float* delta = m->deltas + (size_t)it * NP * 2;
        for (int i = 0; i < NP; ++i) {
            const float dx = delta[2 * i], dy = delta[2 * i + 1];
            const float u = (M[0] * dx + M[1] * dy) + cur[2 * anchor[i]];
            const float v = (M[2] * dx + M[3] * dy) + cur[2 * anchor[i] + 1];
            const double X = (double)u * sx + ox, Y = (double)v * sy + oy;
            const long px = (long)floor(X + 0.5), py = (long)floor(Y + 0.5);
            if (px >= 0 && py >= 0 && px < w && py < h) {
                const uint8_t* p = rgb + ((size_t)py * w + px) * 3;
                pix[i] = (float)(uint8_t)(((unsigned)p[0] + p[1] + p[2]) / 3);
            } else pix[i] = 0;
        }
        for (int t = 0; t < T; ++t) {
            const size_t sb = ((size_t)it * T + t) * n_split;
            int i = 0;
            while (i < n_split) {
                if (pix[m->split_idx1[sb + i]] - pix[m->split_idx2[sb + i]] > m->split_thresh[sb + i]) i = 2 * i + 1;
                else i = 2 * i + 2;
            }
            const float* leaf = m->leaves + (((size_t)it * T + t) * n_leaf + (i - n_split)) * 2 * P;
            for (int k = 0; k < 2 * P; ++k) cur[k] = cur[k] + leaf[k];
        }
    }
    for (int i = 0; i < P; ++i) {
        const double X = (double)cur[2 * i] * sx + ox, Y = (double)cur[2 * i + 1] * sy + oy;
        pts[2 * i] = (int32_t)floor(X + 0.5);
        pts[2 * i + 1] = (int32_t)floor(Y + 0.5);
    }
    free(cur); free(pix);
}
