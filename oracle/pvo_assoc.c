/*
 * pvo_assoc.c -- ORACLE (test infrastructure): tracker/detection association.
 *   reference: pyannote/video/tracking.py:129-134 (_match on dlib.drectangle), :136-182 (_associate, Munkres)
 * munkres: PINNED against the package itself (munkres 1.1.4, setup.py:51 asks for >= 1.0.7; its source is on disk in this
 * container at /opt/conda/lib/python3.9/site-packages/munkres.py and is loaded by tests/test_reference_pins.py), ties
 * included.  The step order below restates Munkres.__step1..6 literally, including step 4's resumed cyclic search
 * (__find_a_zero(i0, j0)), which returns the LAST uncovered zero of the first row that has one.  The optimal cost is
 * additionally pinned against scipy.optimize.linear_sum_assignment.
 */
#include "pvo.h"
#include <stdlib.h>
#include <string.h>

/* dlib.drectangle: width = r-l, height = b-t, empty when l>r or t>b; intersect = (max l, max t, min r, min b) */
static double darea(double l, double t, double r, double b) { return (l > r || t > b) ? 0.0 : (r - l) * (b - t); }

void pvo_overlap_matrix(const double* a, int na, const double* b, int nb, double ratio, double* out)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            const double* p = a + 4 * i;
            const double* q = b + 4 * j;
            const double il = p[0] > q[0] ? p[0] : q[0], it = p[1] > q[1] ? p[1] : q[1];
            const double ir = p[2] < q[2] ? p[2] : q[2], ib = p[3] < q[3] ? p[3] : q[3];
            double ov = darea(il, it, ir, ib);
            if (ov < ratio * darea(p[0], p[1], p[2], p[3]) || ov < ratio * darea(q[0], q[1], q[2], q[3])) ov = 0.0;
            out[(size_t)i * nb + j] = ov;
        }
}

void pvo_munkres(const double* cost, int n, int32_t* row_to_col)
{
    double* C = (double*)malloc(sizeof(double) * n * n);
    memcpy(C, cost, sizeof(double) * n * n);
    char* marked = (char*)calloc((size_t)n * n, 1);
    char* rc = (char*)calloc(n, 1);
    char* cc = (char*)calloc(n, 1);
    int* path = (int*)malloc(sizeof(int) * 4 * n + 8);
    int z0r = 0, z0c = 0, step = 1;
    while (step != 7) {
        if (step == 1) {
            for (int i = 0; i < n; ++i) {
                double mn = C[i * n];
                for (int j = 1; j < n; ++j) if (C[i * n + j] < mn) mn = C[i * n + j];
                for (int j = 0; j < n; ++j) C[i * n + j] -= mn;
            }
            step = 2;
        } else if (step == 2) {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (C[i * n + j] == 0 && !cc[j] && !rc[i]) { marked[i * n + j] = 1; cc[j] = 1; rc[i] = 1; break; }
            memset(rc, 0, n); memset(cc, 0, n);
            step = 3;
        } else if (step == 3) {
            int count = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (marked[i * n + j] == 1 && !cc[j]) { cc[j] = 1; ++count; }
            step = count >= n ? 7 : 4;
        } else if (step == 4) {
            int i0 = 0, j0 = 0;                       /* the search resumes where the previous primed zero was found */
            for (;;) {
                int row = -1, col = -1, i = i0, done = 0;
                while (!done) {
                    int j = j0;
                    do {                              /* no early exit: the last uncovered zero of the row wins */
                        if (C[i * n + j] == 0 && !rc[i] && !cc[j]) { row = i; col = j; done = 1; }
                        j = (j + 1) % n;
                    } while (j != j0);
                    i = (i + 1) % n;
                    if (i == i0) done = 1;
                }
                if (row < 0) { step = 6; break; }
                marked[row * n + col] = 2;
                int star = -1;
                for (int j = 0; j < n; ++j) if (marked[row * n + j] == 1) { star = j; break; }
                if (star >= 0) { rc[row] = 1; cc[star] = 0; i0 = row; j0 = star; }
                else { z0r = row; z0c = col; step = 5; break; }
            }
        } else if (step == 5) {
            int count = 0;
            path[0] = z0r; path[1] = z0c;
            for (;;) {
                int row = -1;
                for (int i = 0; i < n; ++i) if (marked[i * n + path[2 * count + 1]] == 1) { row = i; break; }
                if (row < 0) break;
                ++count; path[2 * count] = row; path[2 * count + 1] = path[2 * (count - 1) + 1];
                int col = -1;
                for (int j = 0; j < n; ++j) if (marked[path[2 * count] * n + j] == 2) { col = j; break; }
                ++count; path[2 * count] = path[2 * (count - 1)]; path[2 * count + 1] = col;
            }
            for (int k = 0; k <= count; ++k) {
                char* mk = &marked[path[2 * k] * n + path[2 * k + 1]];
                *mk = (*mk == 1) ? 0 : 1;
            }
            memset(rc, 0, n); memset(cc, 0, n);
            for (int i = 0; i < n * n; ++i) if (marked[i] == 2) marked[i] = 0;
            step = 3;
        } else if (step == 6) {
            double mn = 0; int have = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (!rc[i] && !cc[j] && (!have || C[i * n + j] < mn)) { mn = C[i * n + j]; have = 1; }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    if (rc[i]) C[i * n + j] += mn;
                    if (!cc[j]) C[i * n + j] -= mn;
                }
            step = 4;
        }
    }
    for (int i = 0; i < n; ++i) {
        row_to_col[i] = -1;
        for (int j = 0; j < n; ++j) if (marked[i * n + j] == 1) { row_to_col[i] = j; break; }
    }
    free(C); free(marked); free(rc); free(cc); free(path);
}
