// shot.hip -- displaced frame difference of the shot boundary detector (reference
// pyannote/video/structure/shot.py:71-99: gray + resize, cv2.calcOpticalFlowFarneback, per-pixel remap, mean |difference|).
// The reference runs OpenCV on tiny images (50 x 88 for 1080p at height = 50) and walks every pixel in a Python loop; here the small gray
// images of ALL frames are made in one launch from the frames resident in HBM, and every consecutive pair is one workgroup that keeps its
// two images, both polynomial expansions, the matrices and the flow in global scratch that never leaves L2 (105 KB per pair).
// Arithmetic order = oracle/pvo_shot.c (which states what is restated from OpenCV and that it is unpinned); float, no contraction
// => the flow and the difference agree with the oracle bit for bit.  The reference's default height (50) gives a single pyramid level for
// any video; an image side of 64 pixels or more brings OpenCV's coarser levels (up to three; round 4), in the same workgroup.
#include "pvf_internal.h"
#include <cmath>

namespace {

__device__ __forceinline__ int gray_of(const uint8_t* p) { return (p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14; }

// ---- shot.py:71-73 for frame blockIdx.y: out[oh][ow]; coefficient tables (source index, 11-bit weights) come from the host
struct Coef { int idx, c0, c1; };
__global__ void __launch_bounds__(256) shot_convert_k(const uint8_t* const* __restrict__ frames, int ih, int iw, const Coef* __restrict__ cy,
                                                      const Coef* __restrict__ cx, uint8_t* __restrict__ out, int oh, int ow)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= oh * ow) return;
    const int y = i / ow, x = i - y * ow;
    const uint8_t* rgb = frames[blockIdx.y];
    const Coef a = cx[x], b = cy[y];
    const int sx1 = min(a.idx + 1, iw - 1), sy1 = min(b.idx + 1, ih - 1);
    const int S0 = gray_of(rgb + ((size_t)b.idx * iw + a.idx) * 3) * a.c0 + gray_of(rgb + ((size_t)b.idx * iw + sx1) * 3) * a.c1;
    const int S1 = gray_of(rgb + ((size_t)sy1 * iw + a.idx) * 3) * a.c0 + gray_of(rgb + ((size_t)sy1 * iw + sx1) * 3) * a.c1;
    out[(size_t)blockIdx.y * oh * ow + i] = (uint8_t)((((b.c0 * (S0 >> 4)) >> 16) + ((b.c1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

__device__ __forceinline__ int reflect101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }

struct Tab { float g[6], xg[6], xxg[6], ig11, ig03, ig33, ig55; };

// Farneback's pyramid (round 4): the coarser levels an image side of 64 pixels or more brings (oracle/pvo_shot.c states what is restated
// from OpenCV).  Level k: size (lh, lw), its Gaussian kernel (sigma = (2^k - 1) / 2; level 0: the fixed 1/4 1/2 1/4 kernel, ksize 0 here).
struct LvTab { int lh, lw, ksize; float k[21]; };
struct FbPlan { int levels; LvTab lv[4]; };

__device__ __forceinline__ void resize_coeff_f(int in, int out, int d, int& idx, float& a0, float& a1)
{
    const double scale = (double)in / out;
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= in - 1) { f = 0; s = in - 1; }
    idx = s; a0 = 1.f - f; a1 = f;
}

// src [ih][iw][cn] -> dst [oh][ow][cn], one thread per output element (the arithmetic of oracle/pvo_shot.c resize_linear_f); `mul`: the flow's x 2
__device__ __forceinline__ void resize_linear_f(const float* src, int ih, int iw, int cn, float* dst, int oh, int ow, bool twice, int tid)
{
    for (int i = tid; i < oh * ow * cn; i += 256) {
        const int c = i % cn, x = (i / cn) % ow, y = i / (cn * ow);
        int sy, sx; float b0, b1, a0, a1;
        resize_coeff_f(ih, oh, y, sy, b0, b1);
        resize_coeff_f(iw, ow, x, sx, a0, a1);
        const int sy1 = min(sy + 1, ih - 1), sx1 = min(sx + 1, iw - 1);
        const float r0 = src[((size_t)sy * iw + sx) * cn + c] * a0 + src[((size_t)sy * iw + sx1) * cn + c] * a1;
        const float r1 = src[((size_t)sy1 * iw + sx) * cn + c] * a0 + src[((size_t)sy1 * iw + sx1) * cn + c] * a1;
        const float v = r0 * b0 + r1 * b1;
        dst[i] = twice ? v * 2.f : v;
    }
}

// One workgroup per pair.  Scratch per pair (floats, px = full-size pixels): F[px], tmp[px], I[2][px], row3[3 px], R[2][5 px], M[5 px],
// cs[5 px], flow[2][2 px] (the two flow buffers take turns from level to level).
__global__ void __launch_bounds__(256) shot_pair_k(const uint8_t* __restrict__ gray, int h, int w, Tab t, FbPlan plan, float* __restrict__ scratch,
                                                   size_t scratch_stride, double* __restrict__ dfd, float* __restrict__ flow_out)
{
    __shared__ int red[256];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int px = h * w;
    const uint8_t* img[2] = {gray + (size_t)pair * px, gray + (size_t)(pair + 1) * px};
    float* base = scratch + (size_t)pair * scratch_stride;
    float* F = base;
    float* tmp = F + px;
    float* I[2] = {tmp + px, tmp + 2 * px};
    float* row3 = tmp + 3 * px;
    float* R[2] = {row3 + 3 * px, row3 + 3 * px + 5 * px};
    float* M = R[1] + 5 * px;
    float* cs = M + 5 * px;
    float* fl = cs + 5 * px;
    float* fl_prev = fl + 2 * px;
    int plh = 0, plw = 0;
    for (int lk = plan.levels; lk >= 0; --lk) {
        const int lh = plan.lv[lk].lh, lw = plan.lv[lk].lw, lpx = lh * lw;
        // ---- the flow this level starts from
        if (lk == plan.levels) { for (int i = tid; i < 2 * lpx; i += 256) fl[i] = 0.f; }
        else resize_linear_f(fl_prev, plh, plw, 2, fl, lh, lw, true, tid);
        __syncthreads();
        // ---- the two level images
        for (int s = 0; s < 2; ++s) {
            if (lk == 0) {
                // float image + 3 x 3 blur (fixed kernel), rows then columns
                for (int i = tid; i < px; i += 256) {
                    const int y = i / w, x = i - y * w;
                    const float a = (float)img[s][y * w + reflect101(x - 1, w)], b = (float)img[s][i], c = (float)img[s][y * w + reflect101(x + 1, w)];
                    tmp[i] = (a * 0.25f + b * 0.5f) + c * 0.25f;
                }
                __syncthreads();
                for (int i = tid; i < px; i += 256) {
                    const int y = i / w, x = i - y * w;
                    const float a = tmp[reflect101(y - 1, h) * w + x], b = tmp[i], c = tmp[reflect101(y + 1, h) * w + x];
                    I[s][i] = (a * 0.25f + b * 0.5f) + c * 0.25f;
                }
                __syncthreads();
            } else {
                // Gaussian blur of the FULL-size float image (rows: taps left to right; columns: centre, then symmetric pairs), then resize
                const int n = plan.lv[lk].ksize, r = n / 2;
                const float* kk = plan.lv[lk].k;
                float* B = M;                                   // (free until the matrices are formed)
                for (int i = tid; i < px; i += 256) F[i] = (float)img[s][i];
                __syncthreads();
                for (int i = tid; i < px; i += 256) {
                    const int y = i / w, x = i - y * w;
                    float sacc = kk[0] * F[y * w + reflect101(x - r, w)];
                    for (int j = 1; j < n; ++j) sacc = sacc + kk[j] * F[y * w + reflect101(x - r + j, w)];
                    tmp[i] = sacc;
                }
                __syncthreads();
                for (int i = tid; i < px; i += 256) {
                    const int y = i / w, x = i - y * w;
                    float sacc = kk[r] * tmp[i];
                    for (int d = 1; d <= r; ++d) sacc = sacc + kk[r + d] * (tmp[reflect101(y + d, h) * w + x] + tmp[reflect101(y - d, h) * w + x]);
                    B[i] = sacc;
                }
                __syncthreads();
                resize_linear_f(B, h, w, 1, I[s], lh, lw, false, tid);
                __syncthreads();
            }
            // ---- polynomial expansion
            const float* src = I[s];
            for (int i = tid; i < lpx; i += 256) {
                const int y = i / lw, x = i - y * lw;
                float s0 = src[i] * t.g[0], s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 1; k <= 5; ++k) {
                    const float p = src[min(y + k, lh - 1) * lw + x], m = src[max(y - k, 0) * lw + x];
                    s0 = s0 + t.g[k] * (p + m);
                    s1 = s1 + t.xg[k] * (p - m);
                    s2 = s2 + t.xxg[k] * (p + m);
                }
                row3[i * 3] = s0; row3[i * 3 + 1] = s1; row3[i * 3 + 2] = s2;
            }
            __syncthreads();
            for (int i = tid; i < lpx; i += 256) {
                const int y = i / lw, x = i - y * lw;
                const float* row = row3 + (size_t)y * lw * 3;
                float b1 = row[x * 3] * t.g[0], b2 = 0.f, b3 = row[x * 3 + 1] * t.g[0], b4 = 0.f, b5 = row[x * 3 + 2] * t.g[0], b6 = 0.f;
#pragma unroll
                for (int k = 1; k <= 5; ++k) {
                    const float* rp = row + min(x + k, lw - 1) * 3;
                    const float* rm = row + max(x - k, 0) * 3;
                    const float tg = rp[0] + rm[0];
                    b1 = b1 + tg * t.g[k];
                    b4 = b4 + tg * t.xxg[k];
                    b2 = b2 + (rp[0] - rm[0]) * t.xg[k];
                    b3 = b3 + (rp[1] + rm[1]) * t.g[k];
                    b6 = b6 + (rp[1] - rm[1]) * t.xg[k];
                    b5 = b5 + (rp[2] + rm[2]) * t.g[k];
                }
                float* d = R[s] + (size_t)i * 5;
                d[1] = b2 * t.ig11;
                d[0] = b3 * t.ig11;
                d[3] = b1 * t.ig03 + b4 * t.ig33;
                d[2] = b1 * t.ig03 + b5 * t.ig33;
                d[4] = b6 * t.ig55;
            }
            __syncthreads();
        }
        auto update_matrices = [&]() {
            for (int i = tid; i < lpx; i += 256) {
                const int y = i / lw, x = i - y * lw;
                const float dx = fl[i * 2], dy = fl[i * 2 + 1];
                float fx = (float)x + dx, fy = (float)y + dy;
                const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
                const float* r0 = R[0] + (size_t)i * 5;
                float r2, r3, r4, r5, r6;
                fx -= (float)x1; fy -= (float)y1;
                if ((unsigned)x1 < (unsigned)(lw - 1) && (unsigned)y1 < (unsigned)(lh - 1)) {
                    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                    const float* p = R[1] + ((size_t)y1 * lw + x1) * 5;
                    const float* q = p + (size_t)lw * 5;
                    r2 = ((a00 * p[0] + a01 * p[5]) + a10 * q[0]) + a11 * q[5];
                    r3 = ((a00 * p[1] + a01 * p[6]) + a10 * q[1]) + a11 * q[6];
                    r4 = ((a00 * p[2] + a01 * p[7]) + a10 * q[2]) + a11 * q[7];
                    r5 = ((a00 * p[3] + a01 * p[8]) + a10 * q[3]) + a11 * q[8];
                    r6 = ((a00 * p[4] + a01 * p[9]) + a10 * q[4]) + a11 * q[9];
                    r4 = (r0[2] + r4) * 0.5f;
                    r5 = (r0[3] + r5) * 0.5f;
                    r6 = (r0[4] + r6) * 0.25f;
                } else {
                    r2 = r3 = 0.f;
                    r4 = r0[2]; r5 = r0[3]; r6 = r0[4] * 0.5f;
                }
                r2 = (r0[0] - r2) * 0.5f;
                r3 = (r0[1] - r3) * 0.5f;
                r2 = r2 + (r4 * dy + r6 * dx);
                r3 = r3 + (r6 * dy + r5 * dx);
                if ((unsigned)(x - 5) >= (unsigned)(lw - 10) || (unsigned)(y - 5) >= (unsigned)(lh - 10)) {
                    const float bd[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
                    const float scale = (x < 5 ? bd[x] : 1.f) * (x >= lw - 5 ? bd[lw - x - 1] : 1.f) * (y < 5 ? bd[y] : 1.f) * (y >= lh - 5 ? bd[lh - y - 1] : 1.f);
                    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
                }
                float* m = M + (size_t)i * 5;
                m[0] = r4 * r4 + r6 * r6;
                m[1] = (r4 + r5) * r6;
                m[2] = r5 * r5 + r6 * r6;
                m[3] = r4 * r2 + r6 * r3;
                m[4] = r6 * r2 + r5 * r3;
            }
            __syncthreads();
        };
        update_matrices();
        for (int it = 0; it < 3; ++it) {
            for (int i = tid; i < lpx; i += 256) {
                const int y = i / lw, x = i - y * lw;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    float sacc = 0.f;
                    for (int d = -7; d <= 7; ++d) sacc = sacc + M[((size_t)y * lw + min(max(x + d, 0), lw - 1)) * 5 + k];
                    cs[(size_t)i * 5 + k] = sacc;
                }
            }
            __syncthreads();
            for (int i = tid; i < lpx; i += 256) {
                const int y = i / lw, x = i - y * lw;
                float v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    float sacc = 0.f;
                    for (int d = -7; d <= 7; ++d) sacc = sacc + cs[((size_t)min(max(y + d, 0), lh - 1) * lw + x) * 5 + k];
                    v[k] = sacc * (1.f / 225.f);
                }
                const float idet = 1.f / ((v[0] * v[2] - v[1] * v[1]) + 1e-3f);
                fl[i * 2] = (v[0] * v[4] - v[1] * v[3]) * idet;
                fl[i * 2 + 1] = (v[2] * v[3] - v[1] * v[4]) * idet;
            }
            __syncthreads();
            if (it < 2) update_matrices();
        }
        float* sw = fl; fl = fl_prev; fl_prev = sw;
        plh = lh; plw = lw;
    }
    const float* flow = fl_prev;                     // level 0's result
    // ---- shot.py:89-99: `dy, dx = flow[y, x]`; reconstruct[y, x] = current[int(clamp(y + dy)), int(clamp(x + dx))]; mean |previous - reconstruct|
    int part = 0;
    for (int i = tid; i < px; i += 256) {
        const int y = i / w, x = i - y * w;
        const float dy = flow[i * 2], dx = flow[i * 2 + 1];          // float32 sums, like NumPy >= 2 evaluates `x + dx` (oracle/pvo_shot.c)
        float fx = (float)x + dx, fy = (float)y + dy;
        if (fx > (float)(w - 1)) fx = (float)(w - 1);
        if (fx < 0) fx = 0;
        if (fy > (float)(h - 1)) fy = (float)(h - 1);
        if (fy < 0) fy = 0;
        const int d = (int)img[0][i] - (int)img[1][(int)fy * w + (int)fx];
        part += d < 0 ? -d : d;
        if (flow_out) { flow_out[((size_t)pair * px + i) * 2] = flow[i * 2]; flow_out[((size_t)pair * px + i) * 2 + 1] = flow[i * 2 + 1]; }
    }
    red[tid] = part;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) dfd[pair] = (double)red[0] / (double)px;       // integer sum: exact in any order
}

// the plan of the pyramid for an oh x ow image, with the oracle's arithmetic (oracle/pvo_shot.c: pvo_farneback)
FbPlan farneback_plan(int oh, int ow)
{
    FbPlan p;
    memset(&p, 0, sizeof p);
    int k = 0;
    double scale = 1;
    for (; k < 3; ++k) {
        scale *= 0.5;
        if (ow * scale < 32 || oh * scale < 32) break;
    }
    p.levels = k;
    for (int lk = 0; lk <= p.levels; ++lk) {
        double sc = 1;
        for (int i = 0; i < lk; ++i) sc *= 0.5;
        const double sigma = (1. / sc - 1) * 0.5;
        int n = (int)std::nearbyint(sigma * 5) | 1;
        if (n < 3) n = 3;
        LvTab& L = p.lv[lk];
        L.lw = (int)std::nearbyint(ow * sc); L.lh = (int)std::nearbyint(oh * sc);
        L.ksize = lk == 0 ? 0 : n;
        if (lk > 0) {
            PVF_REQUIRE(n <= 21, "shot: smoothing kernel larger than planned");
            const double s2 = -0.5 / (sigma * sigma);
            double sum = 0;
            for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; L.k[i] = (float)std::exp(s2 * x * x); sum += L.k[i]; }
            sum = 1. / sum;
            for (int i = 0; i < n; ++i) L.k[i] = (float)(L.k[i] * sum);
        }
    }
    return p;
}

void cv_coeffs(int in, int out, std::vector<Coef>& c)
{
    c.resize(out);
    const double scale = (double)in / out;
    for (int d = 0; d < out; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= in - 1) { f = 0; s = in - 1; }
        c[d].idx = s;
        c[d].c0 = (int)(short)nearbyintf((1.f - f) * 2048.f);
        c[d].c1 = (int)(short)nearbyintf(f * 2048.f);
    }
}

}  // namespace

// dfd[i] = displaced frame difference between frames i and i + 1 (n - 1 values); optional gray images [n][oh][ow] and flows [n-1][oh][ow][2]
void shot_dfd(Ctx* c, const std::vector<Frame>& frames, int ow, int oh, const float* tables22, double* dfd, uint8_t* gray_out, float* flow_out)
{
    const int n = (int)frames.size();
    PVF_REQUIRE(n >= 1 && ow >= 12 && oh >= 12, "shot: at least one frame and a small image of 12 x 12 or more");
    const int ih = frames[0].h, iw = frames[0].w;
    for (const Frame& f : frames) PVF_REQUIRE(f.h == ih && f.w == iw, "shot: frames of one size");
    const int px = oh * ow;
    std::vector<Coef> cy, cx;
    cv_coeffs(ih, oh, cy);
    cv_coeffs(iw, ow, cx);
    const size_t stride = (size_t)px * (1 + 1 + 2 + 3 + 10 + 5 + 5 + 4);
    const size_t coef_bytes = ((size_t)(oh + ow) * sizeof(Coef) + 255) / 256 * 256, ptr_bytes = ((size_t)n * sizeof(void*) + 255) / 256 * 256;
    const size_t gray_bytes = ((size_t)n * px + 255) / 256 * 256, dfd_bytes = ((size_t)n * sizeof(double) + 255) / 256 * 256;
    const size_t flow_bytes = flow_out ? (size_t)(n - 1) * px * 2 * sizeof(float) : 0;
    c->s_misc.ensure(coef_bytes + ptr_bytes + gray_bytes + dfd_bytes + flow_bytes + (size_t)std::max(n - 1, 1) * stride * sizeof(float) + 256);
    uint8_t* q = c->s_misc.as<uint8_t>();
    Coef* d_coef = reinterpret_cast<Coef*>(q); q += coef_bytes;
    const uint8_t** d_ptr = reinterpret_cast<const uint8_t**>(q); q += ptr_bytes;
    uint8_t* d_gray = q; q += gray_bytes;
    double* d_dfd = reinterpret_cast<double*>(q); q += dfd_bytes;
    float* d_flow = flow_out ? reinterpret_cast<float*>(q) : nullptr; q += flow_bytes;
    float* d_scratch = reinterpret_cast<float*>(q);
    std::vector<Coef> both(cy);
    both.insert(both.end(), cx.begin(), cx.end());
    std::vector<const uint8_t*> ptrs(n);
    for (int i = 0; i < n; ++i) ptrs[i] = frames[i].d;
    HIP_CHECK(hipMemcpyAsync(d_coef, both.data(), both.size() * sizeof(Coef), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipMemcpyAsync(d_ptr, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));               // (host vectors go out of scope)
    ProfScope ps(c, "shot");
    hipLaunchKernelGGL(shot_convert_k, dim3((px + 255) / 256, n), dim3(256), 0, c->stream, d_ptr, ih, iw, d_coef, d_coef + oh, d_gray, oh, ow);
    if (n > 1) {
        Tab t;
        memcpy(&t, tables22, sizeof t);
        static_assert(sizeof(Tab) == 22 * sizeof(float), "22 table floats");
        const FbPlan plan = farneback_plan(oh, ow);
        hipLaunchKernelGGL(shot_pair_k, dim3(n - 1), dim3(256), 0, c->stream, d_gray, oh, ow, t, plan, d_scratch, stride, d_dfd, d_flow);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(dfd, d_dfd, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (flow_out) HIP_CHECK(hipMemcpyAsync(flow_out, d_flow, flow_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    if (gray_out) HIP_CHECK(hipMemcpyAsync(gray_out, d_gray, (size_t)n * px, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
