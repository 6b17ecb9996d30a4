// cluster.hip -- K10/K11: track-pair mean Euclidean distances and average-linkage agglomeration
// (reference pyannote/video/face/clustering.py:100-112 pdist + block means, :116-119 merged similarity, :138-141 stop at 0.6).
// float64 like scipy/numpy in the reference; the N x N matrix is never materialised:
//   S[a][j] = sum_{b in track j} ||x_a - x_b||   (one lane per (row, track), X kept dimension-major so lanes coalesce)
//   D[i][j] = sum_{a in track i} S[a][j] / (n_i n_j)
// HAC keeps D in HBM with cached row minima; a merge costs O(T) plus re-scans of the rows whose minimum died.
#include "pvf_internal.h"
#include <cmath>

__global__ void __launch_bounds__(256) transpose_k(const double* __restrict__ X, int N, int dim, double* __restrict__ Xt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * dim) return;
    const int a = (int)(i / dim), k = (int)(i % dim);
    Xt[(size_t)k * N + a] = X[i];
}

// one block per row a: lanes sweep all rows b (dimension-major X => coalesced), distances of a 2048-row chunk sit in LDS,
// then one lane per track continues that track's running sum over the chunk in row order (a single sequential chain).
#define PD_CHUNK 2048
__global__ void __launch_bounds__(256) row_track_sums_k(const double* __restrict__ X, const double* __restrict__ Xt, int N, int dim,
                                                        const int32_t* __restrict__ row_start, int T, double* __restrict__ S, int a0)
{
    extern __shared__ __attribute__((aligned(16))) double sm[]; // xa[dim] + dist[PD_CHUNK]
    double* xa = sm;
    double* dist = sm + dim;
    const int a = a0 + blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < dim; k += 256) xa[k] = X[(size_t)a * dim + k];
    for (int j = tid; j < T; j += 256) S[(size_t)a * T + j] = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < N; c0 += PD_CHUNK) {
        const int c1 = min(c0 + PD_CHUNK, N);
        for (int b = c0 + tid; b < c1; b += 256) {
            double s = 0;
            for (int k = 0; k < dim; ++k) { const double d = xa[k] - Xt[(size_t)k * N + b]; s += d * d; }
            dist[b - c0] = sqrt(s);
        }
        __syncthreads();
        for (int j = tid; j < T; j += 256) {
            const int b0 = max(row_start[j], c0), b1 = min(row_start[j + 1], c1);
            if (b0 < b1) {
                double run = S[(size_t)a * T + j];
                for (int b = b0; b < b1; ++b) run += dist[b - c0];
                S[(size_t)a * T + j] = run;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) track_pair_mean_k(const double* __restrict__ S, const int32_t* __restrict__ row_start, int T,
                                                         double* __restrict__ D, int t0)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t0 + blockIdx.y;
    if (j >= T) return;
    if (i == j) { D[(size_t)i * T + j] = 0.0; return; }
    double sum = 0;
    for (int a = row_start[i]; a < row_start[i + 1]; ++a) sum += S[(size_t)a * T + j];
    const double cnt = (double)(row_start[i + 1] - row_start[i]) * (double)(row_start[j + 1] - row_start[j]);
    D[(size_t)i * T + j] = sum / cnt;
}

// D rows of the tracks [t0, t1) only (all columns): the unit of work when several GPUs split the pairwise distances of one
// global clustering -- a track's rows all live in one contiguous block, so every entry of D is still produced by one sequential
// chain, the same one the single-GPU call runs.  h_D / d_D_keep address the full T x T matrix (rows outside the range untouched).
void pair_mean_dist_dev(Ctx* c, const double* X, int N, int dim, const int32_t* row_start, int T, double* h_D, double** d_D_keep,
                        int t0, int t1)
{
    if (t1 < 0) t1 = T;
    PVF_REQUIRE(0 <= t0 && t0 <= t1 && t1 <= T, "pair_mean_dist: bad track range");
    PVF_REQUIRE(N > 0 && T > 0 && dim > 0 && dim <= 4096, "pair_mean_dist: bad sizes");
    PVF_REQUIRE(row_start[0] == 0 && row_start[T] == N, "pair_mean_dist: row_start must cover [0, N)");
    const size_t xb = (size_t)N * dim * sizeof(double);
    const size_t sb = (size_t)N * T * sizeof(double), db = (size_t)T * T * sizeof(double), rb = (size_t)(T + 1) * sizeof(int32_t);
    c->s_clu0.ensure(2 * xb + sb + rb + 512);
    c->s_clu1.ensure(db + (size_t)T * 64 + 4096);
    uint8_t* p = c->s_clu0.as<uint8_t>();
    double* dX = reinterpret_cast<double*>(p); p += xb;
    double* dXt = reinterpret_cast<double*>(p); p += xb;
    double* dS = reinterpret_cast<double*>(p); p += sb;
    int32_t* dR = reinterpret_cast<int32_t*>(p);
    double* dD = c->s_clu1.as<double>();
    HIP_CHECK(hipMemcpyAsync(dX, X, xb, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(dR, row_start, rb, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, "pdist");
        hipLaunchKernelGGL(transpose_k, dim3((unsigned)(((size_t)N * dim + 255) / 256)), dim3(256), 0, c->stream, dX, N, dim, dXt);
        const int a0 = row_start[t0], a1 = row_start[t1];
        if (a1 > a0)
            hipLaunchKernelGGL(row_track_sums_k, dim3(a1 - a0), dim3(256), (dim + PD_CHUNK) * sizeof(double), c->stream, dX, dXt, N, dim, dR, T, dS, a0);
        if (t1 > t0)
            hipLaunchKernelGGL(track_pair_mean_k, dim3((T + 255) / 256, t1 - t0), dim3(256), 0, c->stream, dS, dR, T, dD, t0);
    }
    HIP_CHECK(hipGetLastError());
    if (h_D && t1 > t0)
        HIP_CHECK(hipMemcpyAsync(h_D + (size_t)t0 * T, dD + (size_t)t0 * T, (size_t)(t1 - t0) * T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (d_D_keep) *d_D_keep = dD;
}

// ---------------------------------------------------------------------------------------------------
struct HacState {
    double* D; int T;
    double* rmin; int* rarg; int* alive; int* dirty; double* size;
    double* best;   // [0]=bi [1]=bj [2]=dist [3]=done
    double* log;    // [(T-1)][4]
    int* n_merges;
    double threshold;
};

__global__ void __launch_bounds__(256) hac_row_min_k(HacState h, int only_dirty)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (only_dirty && !h.dirty[r]) return;
    double bv = INFINITY; int bi = 0x7fffffff;
    if (h.alive[r]) {
        for (int j = r + 1 + tid; j < h.T; j += 256) {
            if (!h.alive[j]) continue;
            const double v = h.D[(size_t)r * h.T + j];
            if (v < bv) { bv = v; bi = j; }
        }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = sv[tid + off]; const int oi = si[tid + off];
            if (ov < sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { h.rmin[r] = sv[0]; h.rarg[r] = si[0]; h.dirty[r] = 0; }
}

__global__ void __launch_bounds__(1024) hac_argmin_k(HacState h)
{
    __shared__ double sv[1024];
    __shared__ int si[1024];
    const int tid = threadIdx.x;
    if (h.best[3] != 0.0) return;
    double bv = INFINITY; int bi = 0x7fffffff;
    for (int r = tid; r < h.T; r += 1024) {
        const double v = h.rmin[r];
        if (v < bv) { bv = v; bi = r; }   // ascending r per thread => first occurrence kept
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) {
            const double ov = sv[tid + off]; const int oi = si[tid + off];
            if (ov < sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (si[0] == 0x7fffffff || !(sv[0] <= h.threshold)) { h.best[3] = 1.0; return; }
        const int i = si[0], j = h.rarg[i];
        h.best[0] = i; h.best[1] = j; h.best[2] = sv[0];
        const int m = *h.n_merges;
        h.log[4 * m] = i; h.log[4 * m + 1] = j; h.log[4 * m + 2] = sv[0]; h.log[4 * m + 3] = h.size[i] + h.size[j];
        *h.n_merges = m + 1;
    }
}

__global__ void __launch_bounds__(256) hac_merge_k(HacState h)
{
    if (h.best[3] != 0.0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int bi = (int)h.best[0], bj = (int)h.best[1];
    if (k >= h.T) return;
    const double si = h.size[bi], sj = h.size[bj];
    if (k == bi) { h.dirty[bi] = 1; return; }
    if (k == bj) { h.dirty[bj] = 1; return; }
    if (!h.alive[k]) return;
    const double v = (si * h.D[(size_t)bi * h.T + k] + sj * h.D[(size_t)bj * h.T + k]) / (si + sj);
    h.D[(size_t)bi * h.T + k] = v;
    h.D[(size_t)k * h.T + bi] = v;
    if (k < bi) {
        const int ra = h.rarg[k];
        if (ra == bi || ra == bj) h.dirty[k] = 1;
        else if (v < h.rmin[k] || (v == h.rmin[k] && bi < ra)) { h.rmin[k] = v; h.rarg[k] = bi; }
    } else if (k < bj) {
        if (h.rarg[k] == bj) h.dirty[k] = 1;
    }
}

__global__ void hac_finish_merge_k(HacState h)
{
    if (h.best[3] != 0.0) return;
    const int bi = (int)h.best[0], bj = (int)h.best[1];
    h.size[bi] = h.size[bi] + h.size[bj];
    h.alive[bj] = 0;
}

__global__ void __launch_bounds__(256) hac_init_k(HacState h, const int32_t* __restrict__ row_start)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { h.best[0] = h.best[1] = h.best[2] = h.best[3] = 0.0; *h.n_merges = 0; }
    if (k >= h.T) return;
    h.alive[k] = 1; h.dirty[k] = 0; h.size[k] = (double)(row_start[k + 1] - row_start[k]);
}

int hac_dev(Ctx* c, double* d_D, const int32_t* row_start, int T, double threshold, int32_t* labels, double* merge_log)
{
    for (int i = 0; i < T; ++i) labels[i] = i;
    if (T < 2) return 0;
    const size_t need = (size_t)T * (8 + 4 + 4 + 4 + 8) + 4 * 8 + (size_t)T * 4 * 8 + 64 + (size_t)(T + 1) * 4 + 512;
    c->s_misc.ensure(need);
    uint8_t* p = c->s_misc.as<uint8_t>();
    auto take = [&](size_t bytes) { uint8_t* q = p; p += (bytes + 63) / 64 * 64; return q; };
    // note: the bump sizes above leave slack for the 64-byte rounding
    c->s_misc.ensure(need + 64 * 10);
    p = c->s_misc.as<uint8_t>();
    HacState h;
    h.D = d_D; h.T = T; h.threshold = threshold;
    h.rmin = (double*)take((size_t)T * 8);
    h.size = (double*)take((size_t)T * 8);
    h.best = (double*)take(4 * 8);
    h.log = (double*)take((size_t)T * 4 * 8);
    h.rarg = (int*)take((size_t)T * 4);
    h.alive = (int*)take((size_t)T * 4);
    h.dirty = (int*)take((size_t)T * 4);
    h.n_merges = (int*)take(64);
    int32_t* dR = (int32_t*)take((size_t)(T + 1) * 4);
    HIP_CHECK(hipMemcpyAsync(dR, row_start, (size_t)(T + 1) * 4, hipMemcpyHostToDevice, c->stream));
    ProfScope ps(c, "hac");
    hipLaunchKernelGGL(hac_init_k, dim3((T + 255) / 256), dim3(256), 0, c->stream, h, dR);
    hipLaunchKernelGGL(hac_row_min_k, dim3(T), dim3(256), 0, c->stream, h, 0);
    double hbest[4];
    for (int it = 0; it < T - 1; ++it) {
        hipLaunchKernelGGL(hac_argmin_k, dim3(1), dim3(1024), 0, c->stream, h);
        hipLaunchKernelGGL(hac_merge_k, dim3((T + 255) / 256), dim3(256), 0, c->stream, h);
        hipLaunchKernelGGL(hac_finish_merge_k, dim3(1), dim3(1), 0, c->stream, h);
        hipLaunchKernelGGL(hac_row_min_k, dim3(T), dim3(256), 0, c->stream, h, 1);
        if ((it & 63) == 63) {
            HIP_CHECK(hipMemcpyAsync(hbest, h.best, sizeof hbest, hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (hbest[3] != 0.0) break;
        }
    }
    HIP_CHECK(hipGetLastError());
    int n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, h.n_merges, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<double> log((size_t)std::max(n, 1) * 4);
    if (n > 0) {
        HIP_CHECK(hipMemcpyAsync(log.data(), h.log, (size_t)n * 4 * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    for (int m = 0; m < n; ++m) {
        const int a = (int)log[4 * m], b = (int)log[4 * m + 1];
        for (int k = 0; k < T; ++k) if (labels[k] == b) labels[k] = a;
        if (merge_log) memcpy(merge_log + 4 * m, &log[4 * m], 4 * sizeof(double));
    }
    return n;
}
