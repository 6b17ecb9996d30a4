// ert.hip -- K5: 68-point landmarks by an ensemble of regression trees (replaces dlib.shape_predictor(path)(rgb, rect);
// reference pyannote/video/face/face.py:58,69-70).  One workgroup per face; pixel gathers, tree walks and leaf sums are
// spread over the lanes, every reduction keeps the order stated in oracle/pvo_ert.c so the integer points are bit-exact.
#include "pvf_internal.h"

struct ErtJob { const uint8_t* img; int h, w; int rect[4]; };

__global__ void __launch_bounds__(256) ert_k(const ErtJob* __restrict__ jobs, int n_cascades, int n_trees, int n_parts, int n_pix, int depth,
                                             const float* __restrict__ initial, const int32_t* __restrict__ anchor,
                                             const float* __restrict__ deltas, const int32_t* __restrict__ idx1,
                                             const int32_t* __restrict__ idx2, const float* __restrict__ thresh,
                                             const float* __restrict__ leaves, int32_t* __restrict__ pts)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P2 = 2 * n_parts;
    float* cur = smem;                 // [P2]
    float* pix = cur + ((P2 + 3) & ~3); // [n_pix]
    int* leaf = reinterpret_cast<int*>(pix + ((n_pix + 3) & ~3)); // [n_trees]
    __shared__ float M[4];
    const ErtJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const int n_split = (1 << depth) - 1, n_leaf = 1 << depth;
    const double sx = (double)job.rect[2] - (double)job.rect[0], sy = (double)job.rect[3] - (double)job.rect[1];
    const double ox = job.rect[0], oy = job.rect[1];
    for (int k = tid; k < P2; k += blockDim.x) cur[k] = initial[k];
    for (int it = 0; it < n_cascades; ++it) {
        __syncthreads();
        if (tid == 0) {
            // linear part of the similarity transform initial -> cur (sequential sums over parts, double)
            double mfx = 0, mfy = 0, mtx = 0, mty = 0;
            for (int i = 0; i < n_parts; ++i) { mfx += initial[2 * i]; mfy += initial[2 * i + 1]; mtx += cur[2 * i]; mty += cur[2 * i + 1]; }
            mfx /= n_parts; mfy /= n_parts; mtx /= n_parts; mty /= n_parts;
            double a = 0, b = 0, s = 0;
            for (int i = 0; i < n_parts; ++i) {
                const double fx = initial[2 * i] - mfx, fy = initial[2 * i + 1] - mfy;
                const double tx = cur[2 * i] - mtx, ty = cur[2 * i + 1] - mty;
                a += fx * tx + fy * ty;
                b += fx * ty - fy * tx;
                s += fx * fx + fy * fy;
            }
            const double ca = a / s, cb = b / s;
            M[0] = (float)ca; M[1] = (float)(-cb); M[2] = (float)cb; M[3] = (float)ca;
        }
        __syncthreads();
        const int32_t* an = anchor + (size_t)it * n_pix;
        const float* dl = deltas + (size_t)it * n_pix * 2;
        for (int i = tid; i < n_pix; i += blockDim.x) {
            const float dx = dl[2 * i], dy = dl[2 * i + 1];
            const int ai = an[i];
            const float u = (M[0] * dx + M[1] * dy) + cur[2 * ai];
            const float v = (M[2] * dx + M[3] * dy) + cur[2 * ai + 1];
            const double X = (double)u * sx + ox, Y = (double)v * sy + oy;
            const long px = (long)floor(X + 0.5), py = (long)floor(Y + 0.5);
            float val = 0.0f;
            if (px >= 0 && py >= 0 && px < job.w && py < job.h) {
                const uint8_t* p = job.img + ((size_t)py * job.w + px) * 3;
                val = (float)(((unsigned)p[0] + p[1] + p[2]) / 3);
            }
            pix[i] = val;
        }
        __syncthreads();
        for (int t = tid; t < n_trees; t += blockDim.x) {
            const size_t sb = ((size_t)it * n_trees + t) * n_split;
            int i = 0;
            while (i < n_split) {
                if (pix[idx1[sb + i]] - pix[idx2[sb + i]] > thresh[sb + i]) i = 2 * i + 1;
                else i = 2 * i + 2;
            }
            leaf[t] = i - n_split;
        }
        __syncthreads();
        for (int k = tid; k < P2; k += blockDim.x) {
            float acc = cur[k];
            const float* lb = leaves + (size_t)it * n_trees * n_leaf * P2 + k;
            // the sum is a chain in tree order (the oracle's), but its 500 operands are independent loads out of a 65 MB table (L2 / MALL):
            // twenty of them are in flight per lane before the first is added (round 4: four -- the loop waited for memory latency 125
            // times per cascade and face: 9.4 ms per 8000 faces)
            int t = 0;
            for (; t + 20 <= n_trees; t += 20) {
                float v[20];
#pragma unroll
                for (int j = 0; j < 20; ++j) v[j] = lb[((size_t)(t + j) * n_leaf + leaf[t + j]) * P2];
#pragma unroll
                for (int j = 0; j < 20; ++j) acc = acc + v[j];
            }
            for (; t < n_trees; ++t) acc = acc + lb[((size_t)t * n_leaf + leaf[t]) * P2];
            cur[k] = acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < n_parts; i += blockDim.x) {
        const double X = (double)cur[2 * i] * sx + ox, Y = (double)cur[2 * i + 1] * sy + oy;
        pts[((size_t)blockIdx.x * n_parts + i) * 2] = (int32_t)floor(X + 0.5);
        pts[((size_t)blockIdx.x * n_parts + i) * 2 + 1] = (int32_t)floor(Y + 0.5);
    }
}

void ert_run(Ctx* c, const std::vector<Frame>& frames, const pvf_rect_i32* boxes, int n, int32_t* pts)
{
    const ShapeModel& s = c->shape;
    PVF_REQUIRE(s.loaded, "shape predictor not loaded");
    if (n == 0) return;
    std::vector<ErtJob> jobs(n);
    for (int i = 0; i < n; ++i) {
        jobs[i].img = frames[i].d; jobs[i].h = frames[i].h; jobs[i].w = frames[i].w;
        jobs[i].rect[0] = boxes[i].left; jobs[i].rect[1] = boxes[i].top; jobs[i].rect[2] = boxes[i].right; jobs[i].rect[3] = boxes[i].bottom;
    }
    const size_t jb = (size_t)n * sizeof(ErtJob), pb = (size_t)n * s.n_parts * 2 * sizeof(int32_t);
    c->s_misc.ensure(jb + pb + 64);
    uint8_t* d_jobs = c->s_misc.as<uint8_t>();
    int32_t* d_pts = reinterpret_cast<int32_t*>(d_jobs + (jb + 63) / 64 * 64);
    HIP_CHECK(hipMemcpyAsync(d_jobs, jobs.data(), jb, hipMemcpyHostToDevice, c->stream));
    const size_t lds = (size_t)(((2 * s.n_parts + 3) & ~3) + ((s.n_pix + 3) & ~3)) * 4 + (size_t)s.n_trees * 4;
    {
        ProfScope ps(c, "ert");
        hipLaunchKernelGGL(ert_k, dim3(n), dim3(256), lds, c->stream, reinterpret_cast<const ErtJob*>(d_jobs), s.n_cascades, s.n_trees,
                           s.n_parts, s.n_pix, s.depth, s.d_initial, s.d_anchor, s.d_deltas, s.d_idx1, s.d_idx2, s.d_thresh, s.d_leaves,
                           d_pts);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(pts, d_pts, pb, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
